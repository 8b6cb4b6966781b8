#!/bin/bash
# Diagnostic build of the library with the gemm_nt_v5 ablation switches compiled in (LNX_V5_STAGGER bit 16: no epilogue,
# bit 17: two K slices only).  Wrong results by design; load with LNX_LIB_PATH=tools/libv5_ablate.so.
set -e
cd "$(dirname "$0")/../linnaeus_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DV5_ABLATE -c gemm2.hip -o /tmp/gemm2_ablate.o
OBJS=$(ls *.o | grep -v '^gemm2.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemm2_ablate.o -o ../../tools/libv5_ablate.so
