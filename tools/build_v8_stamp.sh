#!/bin/bash
# Diagnostic build of the library with the gemm_nt_v8 phase stamps compiled in; load with LNX_LIB_PATH=tools/libv8_stamp.so.
set -e
cd "$(dirname "$0")/../linnaeus_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DV8_STAMP -c gemm4.hip -o /tmp/gemm4_stamp.o
OBJS=$(ls *.o | grep -v '^gemm4.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/gemm4_stamp.o -o ../../tools/libv8_stamp.so
