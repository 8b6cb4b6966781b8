"""gemm_nt_v9 tile height A/B (round 4): the NT products of xl @224 (128 images), lg @384 (64 images) and sm (128 / 256 images) timed with
256- and 224-row tiles (LNX_NT_V9_ROWS, read per launch; the committed log also has 192-row tiles, NM1 = 2, which are no longer instantiated) and with the height pick_v9_rows() chooses, interleaved in one process.
Prints per shape and form: microseconds, TFLOP/s, tiles and rounds on the CUs a launch may use, and which height "auto" took.
usage (GPU box, repo root, at commit 19087f1 -- the experiment was reverted, LNX_NT_V9_ROWS does nothing at HEAD): python tools/bench_v9_rows.py [xl|lg|sm|sm128 ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from linnaeus_amd import _lib as L


def ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else None)


def time_it(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


SETS = {
    # (M, N, K, form): the products of one RoPE block, forward and data gradients
    "xl": [(25600, 3072, 1024, "bias"), (25600, 1024, 1024, "res"), (25600, 4096, 1024, "fc1d"), (25600, 1024, 4096, "res"),
           (25600, 4096, 1024, "mul"), (25600, 1024, 4096, "plain"), (25600, 1024, 1024, "plain"), (25600, 1024, 3072, "plain"),
           (6784, 6144, 2048, "bias"), (6784, 2048, 2048, "res"), (6784, 8192, 2048, "fc1d"), (6784, 2048, 8192, "res")],
    "lg": [(37120, 2304, 768, "bias"), (37120, 768, 768, "res"), (37120, 3072, 768, "fc1d"), (37120, 768, 3072, "res"),
           (37120, 3072, 768, "mul"), (37120, 768, 3072, "plain"), (37120, 768, 768, "plain"), (37120, 768, 2304, "plain")],
    "sm": [(50944, 1536, 384, "fc1d"), (50944, 1536, 384, "mul"), (12800, 2304, 768, "bias"), (12800, 3072, 768, "fc1d")],
    "sm128": [(25472, 1536, 384, "fc1d"), (25472, 1536, 384, "mul"), (25472, 1152, 384, "bias")],
}


def main():
    sets = sys.argv[1:] or ["xl", "lg", "sm", "sm128"]
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cus = lib.lnx_device_cus()
    os.environ["LNX_NT_V7"] = "0"
    for name in sets:
        print(f"== {name} ==", flush=True)
        for M, N, K, form in SETS[name]:
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
            W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
            b = torch.randn(N, device="cuda", generator=g)
            out_f32 = form == "res"
            out = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
            res = torch.randn(M, N, device="cuda", generator=g) if out_f32 else None
            aux = torch.randn(M, N, device="cuda", generator=g).bfloat16() if form == "mul" else None
            c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if form == "fc1d" else None
            a = L.GemmArgs()
            a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
            a.A, a.lda, a.W, a.ldw = ptr(A), K, ptr(W), K
            a.C, a.ldc, a.out_f32 = ptr(out), N, int(out_f32)
            a.c_map = L.RowMap(0, 0, 0)
            if form in ("bias", "res", "fc1d"):
                a.bias = ptr(b)
            if form == "res":
                a.res, a.ldres = ptr(res), N
            if form == "mul":
                a.act, a.aux, a.ldaux = L.ACT_MUL_AUX, ptr(aux), N
            if form == "fc1d":
                a.act, a.c2, a.ldc2 = L.ACT_GELU_D, ptr(c2), N

            def run():
                L.check(lib.lnx_gemm_nt(C.byref(a), st), "lnx_gemm_nt")

            line = f"M={M:6d} N={N:5d} K={K:5d} {form:5s}"
            outs = {}
            for rows in ("256", "224", "auto"):
                if rows == "auto":
                    os.environ.pop("LNX_NT_V9_ROWS", None)
                else:
                    os.environ["LNX_NT_V9_ROWS"] = rows
                t = time_it(run)
                kind, took = lib.lnx_last_nt_kernel(), lib.lnx_last_nt_tile_rows()
                outs[rows] = out.float().clone() if M * N < 60e6 else None
                tag = f"v{kind}" + (f"/{took}" if kind == L.NT_KERNEL_V9 else "")
                if rows != "auto" and kind == L.NT_KERNEL_V9:
                    tiles = -(-M // int(rows)) * (N // 256)
                    tag += f" {tiles}t/{-(-tiles // cus)}r"
                line += f" | {rows}: {t * 1e6:7.1f} us {2.0 * M * N * K / t * 1e-12:6.0f} TF ({tag})"
            if outs["256"] is not None:
                for r in ("224",):
                    d = (outs[r] - outs["256"]).abs().max().item()
                    line += f" |d{r}|={d:.1e}"
            print(line, flush=True)
    os.environ.pop("LNX_NT_V9_ROWS", None)


if __name__ == "__main__":
    main()
