#!/usr/bin/env python3
"""Turn gpurun_out/prof_round/ (tools/profile_round.sh) into the summaries committed under profiles/.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average), LNX_WGRAD_STREAM=0: each kernel alone
  profiles/<tag>_overlap_kernel_stats.csv   the same of the default run (weight-gradient stream on: durations of kernels that ran side by side overlap)
  profiles/<tag>_hbm_traffic.csv    per kernel: launches/step, HBM read and write bytes per launch from the PMC passes
                                    (FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 B; WRITE_SIZE as read;
                                    both counters are in KiB -- MI355X_MICROARCH.md, HBM section)
  profiles/<tag>_gemm_nt_traffic.json   the dominant kernel's bytes per launch; bench.py reports it as roofline.traffic

usage: summarise_profile.py <tag> [steps_traced]
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_round")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "trace", "r_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace_overlap", "r_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "trace_overlap", "r_kernel_stats.csv"), os.path.join(dst, f"{tag}_overlap_kernel_stats.csv"))


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("lnxg::", "")
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:
        n = int(m.group(1))
        rest = name[m.end():]
        name = rest[:n] + "<" + rest[n:] + ">"
    return name[:90]


def per_kernel(sub, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(os.path.join(src, sub, "r_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel("fetch", "FETCH_SIZE")
write = per_kernel("write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1])):
    n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
    rd = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024 / max(fetch.get(k, [1, 0])[0], 1)
    wr = write.get(k, [0, 0.0])[1] * 1024 / max(write.get(k, [1, 0])[0], 1)
    rows.append((k, n / steps, rd, wr))
with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w") as f:
    f.write("kernel,launches_per_step,hbm_read_bytes_per_launch,hbm_write_bytes_per_launch\n")
    for k, n, rd, wr in rows:
        f.write(f"\"{k}\",{n:.2f},{rd:.0f},{wr:.0f}\n")
nt = [(n, rd, wr) for k, n, rd, wr in rows if k.startswith(("gemm_nt_v2_kernel", "gemm_nt_v4_kernel", "gemm_nt_v7_kernel", "gemm_nt_v9_kernel"))]
tot_n = sum(n for n, _, _ in nt)
summary = {"kernel": "gemm_nt_v2 / v4 / v7 / v9 kernels (the pipelined bf16 NT GEMM, all tile shapes and epilogue forms)", "launches_per_step": round(tot_n, 2),
           "hbm_read_bytes_per_launch": round(sum(n * rd for n, rd, _ in nt) / tot_n), "hbm_write_bytes_per_launch": round(sum(n * wr for n, _, wr in nt) / tot_n),
           "source": f"profiles/{tag}_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2)",
           "config": "bench.py --steps 10 --warmup 3, mFormerV1_sm bf16 batch 256"}
summary["bytes_per_launch"] = summary["hbm_read_bytes_per_launch"] + summary["hbm_write_bytes_per_launch"]
with open(os.path.join(dst, f"{tag}_gemm_nt_traffic.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1))
# ---- MFMA utilisation pass (optional): per kernel, matrix-pipe busy cycles / (4 SIMDs x 256 CUs x kernel cycles at the held clock)
mf = os.path.join(src, "mfma", "r_counter_collection.csv")
if os.path.exists(mf):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    with open(mf) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES":
                cnt[k] += 1
    dur = collections.defaultdict(float)
    tr = os.path.join(src, "mfma", "r_kernel_trace.csv")
    if os.path.exists(tr):
        with open(tr) as f:
            for r in csv.DictReader(f):
                dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    with open(os.path.join(dst, f"{tag}_mfma_util.csv"), "w") as f:
        f.write("kernel,launches_per_step,mfma_busy_cycles_per_launch,gui_active_cycles_per_launch_sum8xcd,avg_duration_us,effective_clock_ghz,"
                "mfma_pipe_util\n")
        for k in sorted(acc, key=lambda k: -acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]):
            n = max(cnt[k], 1)
            busy = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n
            gui = acc[k]["GRBM_GUI_ACTIVE"] / n
            d_us = dur[k] / n / 1e3 if dur[k] else 0.0
            clk = gui / 8 / (d_us * 1e3) if d_us else 0.0          # cycles per ns = GHz (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
            util = busy / (4 * 256 * gui / 8) if gui else 0.0       # busy cycles of all matrix pipes / (1024 pipes x kernel cycles)
            f.write(f"\"{k}\",{n / steps:.2f},{busy:.0f},{gui:.0f},{d_us:.1f},{clk:.3f},{util:.4f}\n")
tot_r = sum(n * rd for _, n, rd, _ in rows)
tot_w = sum(n * wr for _, n, _, wr in rows)
print(f"whole step: {tot_r/1e9:.2f} GB read + {tot_w/1e9:.2f} GB written per step")
