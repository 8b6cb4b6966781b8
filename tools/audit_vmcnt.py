#!/usr/bin/env python3
"""Audit: loads that are waited for on the spot.

Compiles every csrc/*.hip to gfx950 assembly and counts, per kernel, the global/buffer loads that are followed within a
few instructions by `s_waitcnt vmcnt(0)` with no other load in between.  A conditional load whose result is merged
with another value (`v = 0; if (ok) v = load(p)`) produces exactly this pattern and serialises every fetch of a loop
(DESIGN.md section 8a).  Epilogues that really need their value at once show up too: read the hits, not the count.

usage: python tools/audit_vmcnt.py [file.hip ...]      (no GPU needed)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

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
    kern, hits = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if kern and re.search(r"\b(global_load|buffer_load)_", l) and "lds" not in l:
            for j in range(i + 1, min(i + 8, len(lines))):
                if re.search(r"(global_load|buffer_load|scratch_load)", lines[j]):
                    break
                if "s_waitcnt vmcnt(0)" in lines[j]:
                    hits[kern] = hits.get(kern, 0) + 1
                    break
    print(os.path.basename(src))
    for k, v in sorted(hits.items(), key=lambda kv: -kv[1]):
        if v >= 2:
            print(f"   {v:3d}  {k[:110]}")
