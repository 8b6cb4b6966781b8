import ctypes as C, sys, torch
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
for name, M, N, K in [("r0.qkv", 50944, 1152, 384), ("r0.fc1", 50944, 1536, 384), ("r0.fc2", 50944, 384, 1536), ("r0.proj", 50944, 384, 384), ("r1.qkv", 13312, 2304, 768), ("r1.fc1", 13312, 3072, 768), ("r1.fc2", 13312, 768, 3072), ("r1.proj", 13312, 768, 768), ("s0.pw1", 802816, 384, 96), ("s0.pw2", 802816, 96, 384), ("s1.pw1", 200704, 768, 192), ("s1.pw2", 200704, 192, 768)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); dY = torch.randn(M, N, device="cuda").bfloat16()
    dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda"); wsb = torch.empty(L.TN_WS_FLOATS, device="cuda")
    res = []
    for splits in [int(v) for v in os.environ.get("TN_SPLITS", "0").split(",")]:
        w = L.WgradArgs(); w.dtype, w.M, w.N, w.K = L.BF16, M, N, K
        w.dY, w.lddy, w.A, w.lda, w.dW, w.lddw, w.db, w.splits = ptr(dY), N, ptr(A), K, ptr(dW), K, ptr(db), splits
        if os.environ.get("TN_WS", "1") == "1": w.ws, w.ws_floats = ptr(wsb), wsb.numel()
        t = time_it(lambda: L.check(L.lib().lnx_gemm_tn(C.byref(w), st()), "tn"))
        res.append(f"s{splits}:{t*1e6:.0f}us/{2.0*M*N*K/t/1e12:.0f}TF")
    print(name, " ".join(res), f"bytes {(M*(N+K)*2)/1e6:.0f} MB -> {(M*(N+K)*2)/t/1e12:.2f} TB/s", flush=True)
