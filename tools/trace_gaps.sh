#!/bin/bash
# Run on the GPU box: plain kernel trace (timestamps, no counters) of a short bench run, then per queue: launches per step, busy time,
# idle time between consecutive kernels of the queue, and the kernels in front of the largest idle gaps.
# usage: trace_gaps.sh <outdir-name> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 "$@" > $O/trace.log 2>&1
echo rc=$?
python3 $R/tools/trace_gaps.py $O/trace/r_kernel_trace.csv | tee $O/gaps.txt
rm -f $O/trace/r_kernel_trace.csv
