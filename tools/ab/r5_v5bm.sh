#!/bin/bash
# the half-height two-workgroups-per-CU tile (tools/experiments/gemm_nt_v5.hip, LNX_V5_BM=128) against the shipped dispatch, sm shapes at 128 and 256 images
cd $GRAFT_REPO_ROOT
echo "== shipped dispatch"; python tools/bench_gemm_forms.py sm128 sm 2>/dev/null
export LNX_LIB_PATH=$GRAFT_REPO_ROOT/tools/liblnx_experiments.so LNX_NT_V5=1
echo "== v5 BM=128"; LNX_V5_BM=128 python tools/bench_gemm_forms.py sm128 sm 2>/dev/null
echo "== v5 BM=256"; python tools/bench_gemm_forms.py sm128 2>/dev/null
