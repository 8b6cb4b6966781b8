cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
export LNX_NT_V3=0
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/p1 -o a -- python3 $R/tools/run_one_gemm.py 50944 1536 384 plain 3 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p2 -o a -- python3 $R/tools/run_one_gemm.py 50944 1536 384 plain 3 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p3 -o a -- python3 $R/tools/run_one_gemm.py 50944 1536 384 plain 3 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/p4 -o a -- python3 $R/tools/run_one_gemm.py 50944 1536 384 plain 3 > $O/p4.log 2>&1
ls $O/*
