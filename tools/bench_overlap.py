"""Do an NT (data-gradient) GEMM and the independent TN (weight-gradient) GEMM of the same layer overlap when issued on two
streams?  Compares back-to-back on one stream with concurrent issue on two."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L
ptr = lambda t: C.c_void_p(t.data_ptr())
M, N, K = 50944, 1536, 384
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * .05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.randn(M, N, device="cuda").bfloat16()
a = L.GemmArgs(); a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
a.A, a.lda, a.W, a.ldw, a.C, a.ldc = ptr(A), K, ptr(W), K, ptr(out), N
a.act, a.aux, a.ldaux = L.ACT_GELU_BWD, ptr(aux), N
dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda"); ws = torch.empty(L.TN_WS_FLOATS, device="cuda")
w = L.WgradArgs(); w.dtype, w.M, w.N, w.K = L.BF16, M, N, K
w.dY, w.lddy, w.A, w.lda, w.dW, w.lddw, w.db = ptr(aux), N, ptr(A), K, ptr(dW), K, ptr(db)
w.ws, w.ws_floats = ptr(ws), ws.numel()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
lib = L.lib()
def run(two):
    st1 = C.c_void_p(s1.cuda_stream); st2 = C.c_void_p((s2 if two else s1).cuda_stream)
    for _ in range(10):
        lib.lnx_gemm_nt(C.byref(a), st1); lib.lnx_gemm_tn(C.byref(w), st2)
for two in (False, True, False, True):
    run(two); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    run(two)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2); e1.record(); torch.cuda.synchronize()
    print("two streams" if two else "one stream ", f"{e0.elapsed_time(e1)/10*1e3:.1f} us per NT+TN pair")
