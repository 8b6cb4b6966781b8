#!/bin/bash
# GPU box: per (kernel, grid, workgroup) average duration over a short bench run -- tells the launches of one kernel apart by shape.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tbg
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $O/trace.log 2>&1
echo rc=$?
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/trace/**/r_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"][:70], r.get("Grid_Size") or "x".join(r.get(k, "?") for k in ("Grid_Size_X","Grid_Size_Y","Grid_Size_Z")), r.get("Workgroup_Size") or r.get("Workgroup_Size_X","?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open("$O/by_grid.txt", "w") as out:
    for (k, g, w), v in rows[:90]:
        out.write(f"{sum(v)/13e3:7.3f} ms/step {len(v)/13:5.1f}/step avg {sum(v)/len(v):8.1f} us grid {g:>9} wg {w:>4}  {k}\n")
print(open("$O/by_grid.txt").read())
PY
rm -rf $O/trace
