#!/bin/bash
# Diagnostic build with the gemm_nt_v5 ablation switches compiled in (LNX_V5_STAGGER bit 16: no epilogue, bit 17: two K slices
# only).  Wrong results by design; load with LNX_LIB_PATH=tools/libv5_ablate.so.
LNX_EXPERIMENTS_OUT="$(dirname "$0")/libv5_ablate.so" exec "$(dirname "$0")/experiments/build.sh" -DV5_ABLATE
