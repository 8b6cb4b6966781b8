#!/bin/bash
# Run on the GPU box: kernel-trace stats of a short bench run; prints the top kernels.  usage: quick_trace.sh <outdir-name> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 "$@" > $O/trace.log 2>&1
echo rc=$?
rm -f $O/trace/r_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/trace/r_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (13 steps):", round(tot / 13e6, 3))
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/13e6:8.3f} ms/step  {int(r["Calls"])/13:6.1f}/step  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:100]}')
PY
