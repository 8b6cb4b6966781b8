"""Build check for gemm_nt_v9 (linnaeus_amd/csrc/gemm5.hip): the first two K iterations of a tile let the previous tile's epilogue stores
stay in flight by COUNT (EpiStores<OUT_F32, F, NM1>::value 16-byte stores per lane; NM1 = 4 / 3: tiles of 256 / 224 rows), so the compiled epilogue must issue at least that many
store instructions and no scratch access.  Compiles the file to assembly and counts, per instantiation:
    global_store_dwordx4 (expected == EpiStores), other global stores (the counter reset only), scratch accesses (expected 0).
usage: python tools/audit_v9_stores.py"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "linnaeus_amd", "csrc", "gemm5.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "g5.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
per_fragment = {(1, 17): 4, (0, 0): 2, (0, 1): 2, (0, 7): 4, (0, 5): 2, (0, 8): 2}  # 16-byte stores per lane and 16-row fragment
bad = seen = 0
for m in re.finditer(r"^(_ZN4lnxg17gemm_nt_v9_kernelILb(\d)ELi(\d+)ELi(\d)EEEvNS_5GemmPE):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    name, o, f, nm1, body = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)
    seen += 1
    x4 = len(re.findall(r"\bglobal_store_dwordx4\b", body))
    other = len(re.findall(r"\bglobal_store_(?!dwordx4)\w+", body))
    scratch = len(re.findall(r"\bscratch_", body))
    want = (4 + nm1) * per_fragment[(o, f)]
    ok = x4 == want and other <= 1 and scratch == 0
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} OUT_F32={o} F={f:2d} rows={2 * (64 + 16 * nm1)}: {x4} 16-byte stores (EpiStores {want}), {other} other stores, {scratch} scratch accesses")
if seen != 12:
    print(f"BAD: {seen} instantiations found, 12 expected")
    bad += 1
sys.exit(1 if bad else 0)
