"""Micro-benchmark of lnx_gemm_nt / lnx_gemm_tn on the mFormerV1_sm shapes (B=256)."""
import ctypes as C
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L

def ptr(t): return C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

shapes = [("s0.pw1", 802816, 384, 96), ("s0.pw2", 802816, 96, 384), ("s1.pw1", 200704, 768, 192), ("s1.pw2", 200704, 192, 768),
          ("r0.qkv", 50944, 1152, 384), ("r0.proj", 50944, 384, 384), ("r0.fc1", 50944, 1536, 384), ("r0.fc2", 50944, 384, 1536),
          ("r1.qkv", 13312, 2304, 768), ("r1.fc1", 13312, 3072, 768), ("r1.fc2", 13312, 768, 3072), ("sq4k", 4096, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "xl":  # mFormerV1_xl @224, B = 128: RoPE stage 3 (M = 128 * 200) and stage 4 (M = 128 * 53)
    shapes = [("xl.qkv", 25600, 3072, 1024), ("xl.proj", 25600, 1024, 1024), ("xl.fc1", 25600, 4096, 1024), ("xl.fc2", 25600, 1024, 4096),
              ("xl4.qkv", 6784, 6144, 2048), ("xl4.fc1", 6784, 8192, 2048), ("xl4.fc2", 6784, 2048, 8192)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a = L.GemmArgs(); a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = ptr(A), K, ptr(W), K, ptr(out), N
    t = time_it(lambda: L.check(L.lib().lnx_gemm_nt(C.byref(a), st()), "nt"))
    t_ref = time_it(lambda: torch.matmul(A, W.T))
    dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    w = L.WgradArgs(); w.dtype, w.M, w.N, w.K = L.BF16, M, N, K
    w.dY, w.lddy, w.A, w.lda, w.dW, w.lddw, w.db = ptr(out), N, ptr(A), K, ptr(dW), K, ptr(db)
    t2 = time_it(lambda: L.check(L.lib().lnx_gemm_tn(C.byref(w), st()), "tn"))
    fl = 2.0 * M * N * K
    byts = 2.0 * (M * K + N * K + M * N)
    print(f"{name:8s} M={M:7d} N={N:5d} K={K:5d}  nt {t*1e6:8.1f}us {fl/t/1e12:7.1f} TF/s {byts/t/1e12:5.2f} TB/s | hipblaslt {t_ref*1e6:8.1f}us {fl/t_ref/1e12:7.1f} TF/s | tn {t2*1e6:8.1f}us {fl/t2/1e12:7.1f} TF/s", flush=True)
