"""Micro-benchmark of lnx_gemm_nt epilogue variants on the RoPE-stage shapes (B=256): the forms plan.cpp launches.

Every case is run under each value of LNX_NT_V5 given on the command line (default "0 1": the 8-wave one-workgroup-per-CU
kernels against the 4-wave two-workgroups-per-CU kernel), interleaved in one process (the switch is read per launch), and
the outputs of the variants are compared with each other and with a torch fp32 product."""
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


modes = sys.argv[1:] or ["base", "v7", "v8"]  # "base" (8-wave kernels), "v7" (persistent deferred-store kernel), "v8" (epilogue waves), "v5[:stagger[:one]]"


def setmode(m):
    name, _, rest = m.partition(":")
    os.environ["LNX_NT_V7"] = "1" if name == "v7" else "0"
    os.environ["LNX_NT_V8"] = "1" if name == "v8" else "0"
    os.environ["LNX_NT_V5"] = "1" if name == "v5" else "0"
    sg, _, one = rest.partition(":")
    os.environ["LNX_V5_STAGGER"] = sg or "0"
    if one:
        os.environ["LNX_V5_ONE_WG"] = "1"
    else:
        os.environ.pop("LNX_V5_ONE_WG", None)


M2, M3 = 50944, 13312
cases = [("r0.qkv   bias            ", M2, 1152, 384, "bias"), ("r0.proj  bias+res f32out ", M2, 384, 384, "res"),
         ("r0.fc1   bias+gelu+c2    ", M2, 1536, 384, "gelu"), ("r0.fc2   bias+res f32out ", M2, 384, 1536, "res"),
         ("r0.dfc2  gelu_bwd(aux)   ", M2, 1536, 384, "gelu_bwd"), ("r0.dfc1  plain           ", M2, 384, 1536, "plain"),
         ("r0.fc1d  bias+gelu+c2=d  ", M2, 1536, 384, "gelu_d"), ("r0.dfc2m mul_aux         ", M2, 1536, 384, "mul_aux"),
         ("r0.dqkv  plain           ", M2, 384, 1152, "plain"), ("r0.dproj plain           ", M2, 384, 384, "plain"),
         ("r1.qkv   bias            ", M3, 2304, 768, "bias"), ("r1.proj  bias+res f32out ", M3, 768, 768, "res"),
         ("r1.fc1   bias+gelu+c2    ", M3, 3072, 768, "gelu"), ("r1.fc2   bias+res f32out ", M3, 768, 3072, "res"),
         ("r1.dfc2  gelu_bwd(aux)   ", M3, 3072, 768, "gelu_bwd"), ("r1.dfc1  plain           ", M3, 768, 3072, "plain"),
         ("r1.dqkv  plain           ", M3, 768, 2304, "plain")]
tot = {m: 0.0 for m in modes}
for name, M, N, K, kind in cases:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    a = L.GemmArgs()
    a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.A, a.lda, a.W, a.ldw = ptr(A), K, ptr(W), K
    keep = []
    ref = A.float() @ W.float().T
    c2 = None
    if kind == "res":
        out = torch.empty(M, N, device="cuda")
        res = torch.randn(M, N, device="cuda")
        a.out_f32, a.res, a.ldres, a.bias = 1, ptr(res), N, ptr(bias)
        rsc = (torch.rand(M // 199 + 1, device="cuda") > 0.2).float() / 0.8  # DropPath multipliers, 199 rows per sample
        a.rowscale, a.rows_per_sample = ptr(rsc), 199
        keep += [res, rsc]
        ref = (ref + bias) * rsc[torch.arange(M, device="cuda") // 199, None] + res
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        if kind not in ("plain", "gelu_bwd", "mul_aux"):  # the data-gradient products have no bias
            a.bias = ptr(bias)
            ref = ref + bias
        if kind == "gelu_d":
            c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            a.act, a.c2, a.ldc2 = L.ACT_GELU_D, ptr(c2), N
            keep += [c2]
            ref = torch.nn.functional.gelu(ref)
        if kind == "mul_aux":
            aux = torch.randn(M, N, device="cuda").bfloat16()
            a.act, a.aux, a.ldaux = L.ACT_MUL_AUX, ptr(aux), N
            keep += [aux]
            ref = ref * aux.float()
        if kind == "gelu":
            c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            a.act, a.c2, a.ldc2 = L.ACT_GELU, ptr(c2), N
            keep += [c2]
            ref = torch.nn.functional.gelu(ref)
        if kind == "gelu_bwd":
            aux = torch.randn(M, N, device="cuda").bfloat16()
            a.act, a.aux, a.ldaux = L.ACT_GELU_BWD, ptr(aux), N
            keep += [aux]
            x = aux.float()
            ref = ref * (0.5 * (1 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327)
    a.C, a.ldc = ptr(out), N

    def run():
        L.check(L.lib().lnx_gemm_nt(C.byref(a), st()), "nt")

    outs, times = {}, {m: [] for m in modes}
    for m in modes:
        setmode(m)
        out.zero_()
        run()
        torch.cuda.synchronize()
        outs[m] = out.float().clone()
    for _ in range(3):  # interleaved rounds
        for m in modes:
            setmode(m)
            times[m].append(time_it(run))
    err = max((outs[m] - ref).abs().max().item() for m in modes) / ref.abs().max().item()
    diff = max((outs[m] - outs[modes[0]]).abs().max().item() for m in modes)
    line = f"{name} M={M:6d} N={N:5d} K={K:5d} "
    for m in modes:
        t = min(times[m])
        tot[m] += t
        line += f" | {m}: {t*1e6:7.1f}us {2.0*M*N*K/t/1e12:6.1f} TF/s"
    print(line + f" | rel err vs fp32 {err:.1e}, max diff between variants {diff:.1e}", flush=True)
    del ref, outs
print("sum: " + "  ".join(f"{m}: {tot[m]*1e6:.1f}us" for m in modes))
