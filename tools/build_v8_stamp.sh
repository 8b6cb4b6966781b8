#!/bin/bash
# Diagnostic build with the gemm_nt_v8 phase stamps compiled in; load with LNX_LIB_PATH=tools/libv8_stamp.so.
LNX_EXPERIMENTS_OUT="$(dirname "$0")/libv8_stamp.so" exec "$(dirname "$0")/experiments/build.sh" -DV8_STAMP
