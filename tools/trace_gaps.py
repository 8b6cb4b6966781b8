"""Idle time between kernels of one training step, from a rocprofv3 --kernel-trace csv (see trace_gaps.sh).  A step = the kernels
between two consecutive adamw_kernel launches (the third-last step of the run)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
i0, i1 = ad[-3], ad[-2]
seg = rows[i0 + 1:i1 + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(seg)} launches")
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = [(int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"]), i) for i in range(len(rs) - 1)]
    pos = sorted(g for g, _ in gaps if g > 0)
    print(f"queue {q}: {len(rs)} launches, busy {busy / 1e6:.3f} ms, idle between its kernels {sum(pos) / 1e6:.3f} ms in {len(pos)} gaps "
          f"(median {pos[len(pos) // 2] / 1e3 if pos else 0:.2f} us, p90 {pos[int(len(pos) * 0.9)] / 1e3 if pos else 0:.2f} us)")
    hist = collections.Counter(min(int(g / 1e3), 20) for g in pos)
    print("   gap histogram (us: count):", " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))
    for g, i in sorted(gaps, reverse=True)[:6]:
        print(f"   {g / 1e3:8.1f} us after {rs[i]['Kernel_Name'][:60]} -> {rs[i + 1]['Kernel_Name'][:60]}")
