"""Weight gradient of the three downsample convs (2x2 patch gather on the A side): pipelined vs register-staged kernel."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L
ptr = lambda t: C.c_void_p(t.data_ptr())
def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
B = 256
for name, H, Cin, Cout in [("ds0", 56, 96, 192), ("ds1", 28, 192, 384), ("ds2", 14, 384, 768)]:
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    M = B * (H // 2) ** 2
    dY = torch.randn(M, Cout, device="cuda").bfloat16()
    dW = torch.zeros(Cout, 4 * Cin, device="cuda"); db = torch.zeros(Cout, device="cuda"); ws = torch.empty(L.TN_WS_FLOATS, device="cuda")
    w = L.WgradArgs(); w.dtype, w.M, w.N, w.K = L.BF16, M, Cout, 4 * Cin
    w.dY, w.lddy, w.A = ptr(dY), Cout, ptr(x)
    w.a_mode, w.Hin, w.Win, w.Cin = L.ADDR_PATCH2, H, H, Cin
    w.dW, w.lddw, w.db, w.k_perm_c = ptr(dW), 4 * Cin, ptr(db), Cin
    w.ws, w.ws_floats = ptr(ws), ws.numel()
    t = time_it(lambda: L.check(L.lib().lnx_gemm_tn(C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tn"))
    print(f"{name} M={M} N={Cout} K={4*Cin}: {t*1e6:.1f} us {2.0*M*Cout*4*Cin/t/1e12:.0f} TF/s", flush=True)
