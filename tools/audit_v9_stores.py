"""Build check for gemm_nt_v9 (linnaeus_amd/csrc/gemm5.hip): the first two K iterations of a tile let the previous tile's epilogue stores
stay in flight by COUNT (EpiStores<OUT_F32, F>::value 16-byte stores per lane), so the compiled epilogue must issue at least that many
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
expect = {(1, 17): 32, (0, 0): 16, (0, 1): 16, (0, 7): 32, (0, 5): 16, (0, 8): 16}
bad = 0
for m in re.finditer(r"^(_ZN4lnxg17gemm_nt_v9_kernelILb(\d)ELi(\d+)EEEvNS_5GemmPE):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
    name, o, f, body = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
    x4 = len(re.findall(r"\bglobal_store_dwordx4\b", body))
    other = len(re.findall(r"\bglobal_store_(?!dwordx4)\w+", body))
    scratch = len(re.findall(r"\bscratch_", body))
    want = expect.get((o, f))
    ok = x4 == want and other <= 1 and scratch == 0
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} OUT_F32={o} F={f:2d}: {x4} 16-byte stores (EpiStores {want}), {other} other stores, {scratch} scratch accesses")
sys.exit(1 if bad else 0)
