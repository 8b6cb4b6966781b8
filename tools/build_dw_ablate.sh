#!/bin/bash
# diagnostic: liblnx_hip.so variants with the depthwise MFMA kernel's loads / stores removed (which side bounds it?)
set -e
cd "$(dirname "$0")/../linnaeus_amd/csrc"
OBJS=$(ls *.o | grep -v dwconv_mfma.o)
for v in NOLOAD NOSTORE NOMFMA NOMFMA2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DDW_$v -c dwconv_mfma.hip -o /tmp/dwconv_mfma_$v.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/dwconv_mfma_$v.o -o ../../tools/libdw_$v.so
done
echo built
