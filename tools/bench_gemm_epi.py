"""Micro-benchmark of lnx_gemm_nt epilogue variants on the RoPE-stage shapes (B=256): the forms plan.cpp launches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L


def ptr(t):
    return C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


M = 50944
cases = [("qkv   bias            ", 1152, 384, "bias"), ("proj  bias+res f32out ", 384, 384, "res"), ("fc1   bias+gelu+c2    ", 1536, 384, "gelu"),
         ("fc2   bias+res f32out ", 384, 1536, "res"), ("dfc2  gelu_bwd(aux)   ", 1536, 384, "gelu_bwd"), ("dfc1  plain           ", 384, 1536, "plain")]
for name, N, K, kind in cases:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.A, a.lda, a.W, a.ldw = ptr(A), K, ptr(W), K
    keep = []
    if kind == "res":
        out = torch.empty(M, N, device="cuda")
        res = torch.randn(M, N, device="cuda")
        a.out_f32, a.res, a.ldres, a.bias = 1, ptr(res), N, ptr(bias)
        keep += [res]
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        if kind != "plain":
            a.bias = ptr(bias)
        if kind == "gelu":
            c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            a.act, a.c2, a.ldc2 = L.ACT_GELU, ptr(c2), N
            keep += [c2]
        if kind == "gelu_bwd":
            aux = torch.randn(M, N, device="cuda").bfloat16()
            a.act, a.aux, a.ldaux = L.ACT_GELU_BWD, ptr(aux), N
            keep += [aux]
    a.C, a.ldc = ptr(out), N
    t = time_it(lambda: L.check(L.lib().lnx_gemm_nt(C.byref(a), st()), "nt"))
    print(f"{name} N={N:5d} K={K:5d}  {t*1e6:8.1f}us {2.0*M*N*K/t/1e12:7.1f} TF/s", flush=True)
