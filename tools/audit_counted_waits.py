"""Build check for the persistent NT kernels' COUNTED waits.

gemm_nt_v9 (csrc/gemm5.hip) lets the previous tile's epilogue stores stay in flight under the next tile's first two K iterations by adding
their COUNT to `s_waitcnt vmcnt(..)` (EpiStores<OUT_F32, F>::value 16-byte stores per lane); gemm_nt_v7 (csrc/gemm3.hip) issues a tile's
NSTORE deferred 16-byte stores in the memory phases of the next tile's first iterations and adds each group's size to that iteration's wait.
Both are only right if the compiled code issues EXACTLY as many store instructions as the source counts: fewer, and a K slice is read
from LDS before it has landed.  This compiles both files to assembly for the architecture the library is built for and checks, per kernel
instantiation: the number of global_store_dwordx4 (v9: EpiStores; v7: 2 x NSTORE = the deferred set + the immediate set of partial / last
tiles), at most one other store (the tile-counter reset) and no scratch access.

    python tools/audit_counted_waits.py [--arch gfx950]        exit code 1 on a mismatch
Also run by __graft_entry__.build() and by tests/test_host_logic.py (CPU only: hipcc cross-compiles)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "linnaeus_amd", "csrc")
F_BIAS, F_C2, F_GELU = 1, 2, 4  # gemm_common.hpp


def _asm(src, hipcc, arch):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", f"--offload-arch={arch}", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
        with open(out) as fh:
            return fh.read()


def _counts(body):
    return (len(re.findall(r"\bglobal_store_dwordx4\b", body)), len(re.findall(r"\bglobal_store_(?!dwordx4)\w+", body)), len(re.findall(r"\bscratch_", body)))


def audit(hipcc=None, arch="gfx950"):
    """-> (ok, report lines)"""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lines, bad, seen = [], 0, 0
    # ---- gemm_nt_v9: template <bool OUT_F32, int F>; EpiStores = 2 * (OUT_F32 ? 16 : 8) + (F & F_C2 ? 16 : 0)
    text = _asm(os.path.join(CSRC, "gemm5.hip"), hipcc, arch)
    for m in re.finditer(r"^(_ZN4lnxg17gemm_nt_v9_kernelILb(\d)ELi(\d+)EEEvNS_5GemmPE):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        o, f, body = int(m.group(2)), int(m.group(3)), m.group(4)
        want = 2 * (16 if o else 8) + (16 if f & F_C2 else 0)
        x4, other, scratch = _counts(body)
        ok = x4 == want and other <= 1 and scratch == 0
        bad += not ok
        seen += 1
        lines.append(f"{'ok ' if ok else 'BAD'} gemm_nt_v9<OUT_F32={o}, F={f}>: {x4} 16-byte stores (EpiStores {want}), {other} other stores, {scratch} scratch accesses")
    # ---- gemm_nt_v7: template <bool OUT_F32, int F, int HEAD, int TAIL>; NSTORE = OUT_F32 ? 16 : (TWO ? 16 : 8), TWO = F has C2 and GELU
    text = _asm(os.path.join(CSRC, "gemm3.hip"), hipcc, arch)
    for m in re.finditer(r"^(_ZN4lnxg17gemm_nt_v7_kernelILb(\d)ELi(\d+)ELi(\d+)ELi(\d+)EEEvNS_5GemmPE):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        o, f, body = int(m.group(2)), int(m.group(3)), m.group(6)
        two = (f & F_C2) and (f & F_GELU)
        want = 2 * (16 if (o or two) else 8)
        x4, other, scratch = _counts(body)
        ok = x4 == want and other <= 1 and scratch == 0
        bad += not ok
        seen += 1
        lines.append(f"{'ok ' if ok else 'BAD'} gemm_nt_v7<OUT_F32={o}, F={f}, {m.group(4)}, {m.group(5)}>: {x4} 16-byte stores (2 x NSTORE = {want}), {other} other stores, {scratch} scratch accesses")
    if seen < 12:
        bad += 1
        lines.append(f"BAD only {seen} kernel instantiations found in the assembly (mangled names changed?)")
    return bad == 0, lines


if __name__ == "__main__":
    arch = sys.argv[sys.argv.index("--arch") + 1] if "--arch" in sys.argv else "gfx950"
    ok, lines = audit(arch=arch)
    print("\n".join(lines))
    sys.exit(0 if ok else 1)
