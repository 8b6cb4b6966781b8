#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + the two PMC passes (HBM read / write bytes) of the default bench.
# Outputs under gpurun_out/prof_round/; tools/summarise_profile.py turns them into the files kept in profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
mkdir -p $O
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
echo rc=$?
rm -f $O/trace/r_kernel_trace.csv  # large; the stats file is what is kept
ls -la $O/*
