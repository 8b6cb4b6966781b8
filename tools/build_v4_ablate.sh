#!/bin/bash
# diagnostic: liblnx_hip.so variant whose 256x256 NT GEMM skips a third of its LDS fragment reads (results are wrong):
#   LNX_LIB_PATH=tools/libv4_ablate.so python tools/bench_gemm.py xl     -- how much of the kernel's time is the LDS port?
set -e
cd "$(dirname "$0")/../linnaeus_amd/csrc"
OBJS=$(ls *.o | grep -v '^gemm2.o$')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DV4_ABLATE_AF1 -c gemm2.hip -o /tmp/gemm2_ablate.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemm2_ablate.o -o ../../tools/libv4_ablate.so
# ... and one without the in-loop LDS-DMA (tools/libv4_nodma.so): how much is the fill path?
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DV4_ABLATE_DMA -c gemm2.hip -o /tmp/gemm2_nodma.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemm2_nodma.o -o ../../tools/libv4_nodma.so
echo built
