#!/bin/bash
# GPU box: memory-side traffic and L2 hit rate of the depthwise kernels (tools/bench_dw.py); separate PMC passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_dw
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python3 $R/tools/bench_dw.py > $O/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python3 $R/tools/bench_dw.py > $O/write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o r -- python3 $R/tools/bench_dw.py > $O/l2.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections
for name in ("fetch", "write", "l2"):
    f = glob.glob("$O/%s/**/r_counter_collection.csv" % name, recursive=True)
    if not f: print(name, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "dwconv" not in k: continue
        agg[(k[:60], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: sum(v) / len(v) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
