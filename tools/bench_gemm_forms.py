"""Per product of the RoPE blocks at the xl / lg (and sm) bench shapes: TFLOP/s of lnx_gemm_nt under the default dispatch with the plan's own epilogue
form, of the SAME kernel family bare (plain epilogue), and of the vendor library (torch.matmul = hipBLASLt) on the same operands -- what each fused
epilogue costs a product in-model, and where the kernels stand against the vendor GEMM.  usage: python tools/bench_gemm_forms.py [xl lg sm]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L  # noqa: E402

NAMES = {1: "v1", 2: "v2", 3: "skinny", 4: "v4", 7: "v7", 9: "v9", 15: "exp"}


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def time_it(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def make(M, N, K, form, rps):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K, a.A, a.lda, a.W, a.ldw = L.BF16, M, N, K, ptr(A), K, ptr(W), K
    keep = [A, W]
    bias = torch.randn(N, device="cuda", generator=g)
    if form == "res_f32":
        out = torch.randn(M, N, device="cuda", generator=g)
        rs = torch.ones(-(-M // rps), device="cuda")
        a.bias, a.res, a.ldres, a.rowscale, a.rows_per_sample, a.out_f32 = ptr(bias), ptr(out), N, ptr(rs), rps, 1
        keep += [rs]
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        if form == "bias":
            a.bias = ptr(bias)
        elif form == "mul_aux":
            aux = torch.randn(M, N, device="cuda", generator=g).bfloat16()
            a.act, a.aux, a.ldaux = L.ACT_MUL_AUX, ptr(aux), N
            keep += [aux]
        elif form == "fc1d":
            c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            a.bias, a.act, a.c2, a.ldc2 = ptr(bias), L.ACT_GELU_D, ptr(c2), N
            keep += [c2]
    a.C, a.ldc = ptr(out), N
    keep += [out, bias]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return (lambda: L.check(L.lib().lnx_gemm_nt(C.byref(a), st), "lnx_gemm_nt")), keep, A, W


CFG = {"xl": (128 * 199, 1024, 4096, 199), "lg": (64 * 580, 768, 3072, 580), "sm": (256 * 199, 384, 1536, 199), "sm128": (128 * 199, 384, 1536, 199)}
for name in (sys.argv[1:] or ["xl", "lg", "sm"]):
    M, C_, hid, rps = CFG[name]
    print(f"# {name}: M = {M}, C = {C_}, hidden = {hid}    TFLOP/s: in-model form | same shape, plain epilogue | vendor (torch.matmul)")
    for prod, N, K, form in (("qkv", 3 * C_, C_, "bias"), ("proj", C_, C_, "res_f32"), ("fc1", hid, C_, "fc1d"), ("fc2", C_, hid, "res_f32"),
                             ("fc2 dgrad", hid, C_, "mul_aux"), ("fc1 dgrad", C_, hid, "plain"), ("proj dgrad", C_, C_, "plain"), ("qkv dgrad", C_, 3 * C_, "plain")):
        fl = 2.0 * M * N * K
        fn, keep, A, W = make(M, N, K, form, rps)
        t_form = time_it(fn)
        kind = NAMES.get(L.lib().lnx_last_nt_kernel(), "?")
        fn0, keep0, _, _ = make(M, N, K, "plain", rps)
        t_plain = time_it(fn0)
        kind0 = NAMES.get(L.lib().lnx_last_nt_kernel(), "?")
        Wt = W.t().contiguous()
        t_vendor = time_it(lambda: torch.matmul(A, Wt))
        print(f"{prod:11s} N={N:5d} K={K:5d} {form:8s} {kind:3s} {fl / t_form / 1e12:7.0f}  ({t_form * 1e6:6.1f} us) | {kind0:3s} {fl / t_plain / 1e12:7.0f} | {fl / t_vendor / 1e12:7.0f}")
        del keep, keep0
