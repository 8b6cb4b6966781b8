#!/bin/bash
# builds gpurun_out/libstamp.so = liblnx_hip.so with convmlp_wgrad.hip compiled -DCW_STAMP (diagnostic only)
set -e
cd "$(dirname "$0")/../linnaeus_amd/csrc"
mkdir -p ../../gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DCW_STAMP -c convmlp_wgrad.hip -o /tmp/convmlp_wgrad_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DDW_STAMP -c dwconv.hip -o /tmp/dwconv_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DDW_STAMP -c dwconv_mfma.hip -o /tmp/dwconv_mfma_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DATT_STAMP -c attention.hip -o /tmp/attention_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DCM_STAMP -c convmlp.hip -o /tmp/convmlp_stamp.o
OBJS=$(ls *.o | grep -v "^convmlp.o" | grep -v convmlp_wgrad.o | grep -v dwconv.o | grep -v dwconv_mfma.o | grep -v attention.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/convmlp_wgrad_stamp.o /tmp/dwconv_stamp.o /tmp/dwconv_mfma_stamp.o /tmp/attention_stamp.o /tmp/convmlp_stamp.o -o ../../tools/libstamp.so
echo built tools/libstamp.so
