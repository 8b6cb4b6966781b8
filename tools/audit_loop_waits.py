#!/usr/bin/env python3
"""Audit: loops that drain the memory counter.

Compiles csrc/*.hip to gfx950 assembly and lists, per kernel, the basic blocks that belong to a loop, contain or follow vector
memory operations and wait with `s_waitcnt vmcnt(0)`.  In a tile loop that prefetches, vmcnt(0) means the prefetch is waited for on
the spot -- usually because some load or store of the loop sits under a branch.  Loops whose every iteration
really needs all of its loads show up too: read the hits.

usage: python tools/audit_loop_waits.py [file.hip ...]      (no GPU needed)
"""
import glob, os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = [os.path.abspath(a) for a in sys.argv[1:]] or sorted(glob.glob(os.path.join(root, "linnaeus_amd", "csrc", "*.hip")))
for src in srcs:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", tmp.name],
                           capture_output=True, text=True, cwd=os.path.dirname(src))
        if r.returncode:
            print(src, "does not compile:", r.stderr[-300:])
            continue
        lines = open(tmp.name).read().split("\n")
    print(os.path.basename(src))
    kern, inloop, stats = None, False, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, inloop = m.group(1), False
            stats[kern] = [0, 0, 0]  # drains in loops, vm loads in loops, vm stores in loops
            continue
        if kern is None:
            continue
        if re.match(r"^\.LBB", l):
            inloop = "Loop" in l
        if "s_endpgm" in l:
            kern = None
            continue
        if inloop:
            if re.search(r"s_waitcnt.*vmcnt\(0\)", l):
                stats[kern][0] += 1
            if re.search(r"\b(global_load|buffer_load|scratch_load)", l):
                stats[kern][1] += 1
            if re.search(r"\b(global_store|buffer_store|scratch_store)", l):
                stats[kern][2] += 1
    for k, (d, nl, ns) in stats.items():
        if d and (nl or ns):
            print(f"  {d:3d} drains, {nl:3d} loads, {ns:3d} stores in loop blocks  {k}")
