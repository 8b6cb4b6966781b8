"""Launch one lnx_gemm_nt shape a few times (target for rocprofv3 --pmc runs).  usage: run_one_gemm.py M N K [kind] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L

M, N, K = (int(v) for v in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "plain"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ptr = lambda t: C.c_void_p(t.data_ptr())
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
else:
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kind == "gelu":
        c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        a.act, a.c2, a.ldc2, a.bias = L.ACT_GELU, ptr(c2), N, ptr(bias)
    if kind == "gelu_bwd":
        aux = torch.randn(M, N, device="cuda").bfloat16()
        a.act, a.aux, a.ldaux = L.ACT_GELU_BWD, ptr(aux), N
a.C, a.ldc = ptr(out), N
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    L.check(L.lib().lnx_gemm_nt(C.byref(a), st), "nt")
torch.cuda.synchronize()
print("done")
