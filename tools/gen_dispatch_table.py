"""The NT dispatch table of DESIGN.md (between the dispatch-table markers), generated from the library's own decision function
lnx_nt_dispatch (gemm2.hip: nt_v2_family) -- no GPU needed.  tests/test_host_logic.py regenerates it and compares with DESIGN.md.

    python tools/gen_dispatch_table.py            print the table
    python tools/gen_dispatch_table.py --write    replace the table in DESIGN.md"""
import ctypes as C
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from linnaeus_amd import _lib as L  # noqa: E402

NAMES = {L.NT_KERNEL_V1: "gemm_nt (128x128)", L.NT_KERNEL_V2: "gemm_nt_v2 (256x128)", L.NT_KERNEL_SKINNY: "gemm_nt_skinny", L.NT_KERNEL_V4: "gemm_nt_v4 (256x256)",
         L.NT_KERNEL_V7: "gemm_nt_v7 (persistent 256x128)", L.NT_KERNEL_V9: "gemm_nt_v9 (persistent 256x256)"}
BEGIN, END = "<!-- dispatch-table-begin -->", "<!-- dispatch-table-end -->"


def query(M, N, K, form, rps=199):
    a = L.GemmArgs()
    one = C.c_void_p(16)
    a.dtype, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = L.BF16, M, N, K, K, K, N
    a.A = a.W = a.C = one
    if form == "bias":
        a.bias = one
    elif form == "res_f32":  # bias + DropPath row scale + fp32 residual, fp32 output (proj / fc2)
        a.bias = a.res = a.rowscale = one
        a.ldres, a.out_f32, a.rows_per_sample = N, 1, rps
    elif form == "mul_aux":  # x saved GELU' factor (fc2 data gradient)
        a.act, a.aux, a.ldaux = L.ACT_MUL_AUX, one, N
    elif form == "fc1d":     # bias + GELU, second output GELU' (fc1 of a training plan)
        a.bias = a.c2 = one
        a.act, a.ldc2 = L.ACT_GELU_D, N
    elif form != "plain":
        raise ValueError(form)
    return L.lib().lnx_nt_dispatch(C.byref(a))


def block_products(C_, hid):
    return [("qkv", 3 * C_, C_, "bias"), ("proj", C_, C_, "res_f32"), ("fc1", hid, C_, "fc1d"), ("fc2", C_, hid, "res_f32"),
            ("fc2 dgrad", hid, C_, "mul_aux"), ("fc1 dgrad", C_, hid, "plain"), ("proj dgrad", C_, C_, "plain"), ("qkv dgrad", C_, 3 * C_, "plain")]


# (stage 4 of sm -- M = 13 312 / 6 656, C = 768: gemm_nt_v4 / v9 / v7 at 256 images, gemm_nt_v2 at 128 -- is left out of the document's table
# for length; query() answers for any shape)
CONFIGS = [("sm @224, 256 img, stage 3", 256 * 199, 384, 1536, 199), ("sm @224, 128 img, stage 3", 128 * 199, 384, 1536, 199),
           ("lg @384, 64 img, stage 3", 64 * 580, 768, 3072, 580), ("xl @224, 128 img, stage 3", 128 * 199, 1024, 4096, 199)]


def table():
    rows = ["| configuration | M | product | N | K | epilogue form | kernel |", "|---|---|---|---|---|---|---|"]
    for name, M, C_, hid, rps in CONFIGS:
        for prod, N, K, form in block_products(C_, hid):
            rows.append(f"| {name} | {M} | {prod} | {N} | {K} | {form} | {NAMES[query(M, N, K, form, rps)]} |")
    return "\n".join(rows)


def parse(text):
    m = re.search(re.escape(BEGIN) + r"\n(.*?)" + re.escape(END), text, re.S)
    return m.group(1).strip() if m else None


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        p = os.path.join(REPO, "DESIGN.md")
        s = open(p).read()
        assert parse(s) is not None, "DESIGN.md has no dispatch-table markers"
        s = re.sub(re.escape(BEGIN) + r"\n.*?" + re.escape(END), lambda m_: BEGIN + "\n" + t + "\n" + END, s, flags=re.S)
        open(p, "w").write(s)
    else:
        print(t)
