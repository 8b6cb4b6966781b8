"""Static audit of the innermost loops of every kernel of a HIP source: branches inside the loop and loads that are waited
for within two instructions (`ds_read ...; s_waitcnt lgkmcnt(0)` / `global_load ...; s_waitcnt vmcnt(0)`), the signature of
a load that ended up under a branch.  Usage: python tools/audit_loops.py linnaeus_amd/csrc/attention.hip [...]"""
import re
import subprocess
import sys
import tempfile

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S"]


def audit(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", tmp.name], check=True, stderr=subprocess.DEVNULL)
        txt = open(tmp.name).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), [l.strip() for l in m.group(2).split("\n")]
        for i, l in enumerate(body):
            mm = re.match(r"(\.LBB\d+_\d+):", l)
            if not mm or ("Inner Loop Header" not in l and not (i + 1 < len(body) and "Inner Loop Header" in body[i + 1])):
                continue
            lab = mm.group(1)
            for j in range(i + 1, len(body)):
                if re.search(r"s_c?branch\w* " + re.escape(lab) + r"$", body[j]):
                    seg = body[i:j]
                    nbr = sum(1 for x in seg if x.startswith("s_cbranch"))
                    nmf = sum(1 for x in seg if "v_mfma" in x)
                    nsc = sum(1 for x in seg if x.startswith("scratch_"))
                    tight = 0
                    for k, x in enumerate(seg):
                        if x.startswith(("ds_read", "global_load", "buffer_load")):
                            if any("s_waitcnt" in y and ("lgkmcnt(0)" in y or "vmcnt(0)" in y) for y in seg[k + 1:k + 3]):
                                tight += 1
                    if len(seg) > 60 and (nbr > 4 or tight > 2 or nsc):
                        print(f"{src.split('/')[-1]:18s} {name[14:80]:66s} loop of {len(seg):4d} instr: {nbr:3d} branches, {nmf:3d} MFMA, {tight:3d} tight waits, {nsc:2d} scratch")
                    break


if __name__ == "__main__":
    for s in sys.argv[1:]:
        audit(s)
