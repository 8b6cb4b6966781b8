"""Conv-stage weight-gradient shapes of mFormerV1_sm (B = 256) under both tile orientations of gemm_tn_v2 (LNX_TN_ORIENT is read
once per process: run as  python tools/bench_tn_orient.py  which re-runs itself for r / c / default)."""
import ctypes as C
import os
import subprocess
import sys

if len(sys.argv) == 1:
    for o in ("", "r", "c"):
        env = dict(os.environ)
        if o:
            env["LNX_TN_ORIENT"] = o
        else:
            env.pop("LNX_TN_ORIENT", None)
        print(f"--- LNX_TN_ORIENT={o or '(default)'}", flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=True)
    sys.exit(0)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L

wsb = torch.empty(L.TN_WS_FLOATS, device="cuda")
for name, M, N, K in (("s0 dW2 = dy^T act ", 802816, 96, 384), ("s0 dW1 = dh^T ln  ", 802816, 384, 96), ("s1 dW2           ", 200704, 192, 768), ("s1 dW1           ", 200704, 768, 192)):
    dY = torch.randn(M, N, device="cuda").bfloat16()
    A = torch.randn(M, K, device="cuda").bfloat16()
    dW = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    a = L.WgradArgs()
    a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
    a.dY, a.lddy, a.A, a.lda = dY.data_ptr(), N, A.data_ptr(), K
    a.dW, a.lddw, a.db = dW.data_ptr(), K, db.data_ptr()
    a.ws, a.ws_floats = wsb.data_ptr(), wsb.numel()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fn():
        L.check(L.lib().lnx_gemm_tn(C.byref(a), st), "tn")

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"{name} M={M} N={N} K={K}: {t * 1e6:7.1f} us  {(M * (N + K) * 2) / t / 1e12:5.2f} TB/s of operands  {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s", flush=True)
