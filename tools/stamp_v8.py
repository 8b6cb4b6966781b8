"""Phase times of one gemm_nt_v8 workgroup (diagnostic build tools/libv8_stamp.so): s_memtime sums of its K-loop wave 0 and its
epilogue wave 4.  Run: LNX_LIB_PATH=tools/libv8_stamp.so python tools/stamp_v8.py [fc1|bias|plain|res|mul_aux]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L

kind = sys.argv[1] if len(sys.argv) > 1 else "fc1"
M, N, K = (50944, 1536, 384) if kind in ("fc1", "mul_aux") else ((50944, 384, 1536) if kind in ("res", "plain") else (50944, 1152, 384))
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
a = L.GemmArgs()
a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
a.A, a.lda, a.W, a.ldw = C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K
keep = []
if kind == "res":
    out = torch.empty(M, N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    a.out_f32, a.res, a.ldres, a.bias = 1, C.c_void_p(res.data_ptr()), N, C.c_void_p(bias.data_ptr())
    keep.append(res)
else:
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kind in ("fc1", "bias"):
        a.bias = C.c_void_p(bias.data_ptr())
    if kind == "fc1":
        c2 = torch.empty_like(out)
        a.act, a.c2, a.ldc2 = L.ACT_GELU_D, C.c_void_p(c2.data_ptr()), N
        keep.append(c2)
    if kind == "mul_aux":
        aux = torch.randn(M, N, device="cuda").bfloat16()
        a.act, a.aux, a.ldaux = L.ACT_MUL_AUX, C.c_void_p(aux.data_ptr()), N
        keep.append(aux)
a.C, a.ldc = C.c_void_p(out.data_ptr()), N
stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
os.environ["LNX_NT_V8"] = "1"
os.environ["LNX_V8_STAMPS"] = str(stamps.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    L.check(L.lib().lnx_gemm_nt(C.byref(a), st), "nt")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.check(L.lib().lnx_gemm_nt(C.byref(a), st), "nt")
e1.record()
torch.cuda.synchronize()
s = stamps.cpu().tolist()
tiles = ((M + 255) // 256) * ((N + 127) // 128)
print(f"{kind}: M={M} N={N} K={K}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch, {tiles} tiles = {tiles / 256:.2f} per workgroup, {K // 32} K iterations per tile")
kn = ["MFMAs + fragment reads + LDS-DMA issue", "wait for own pieces of slice k+2", "wait for fragments", "iteration barrier", "-", "barrier: epilogue waves done", "image write + barrier", "-"]
en = ["epilogue units", "iteration barriers", "hand-over barriers", "last image", "-", "-", "-", "-"]
tk, te = sum(s[:8]), sum(s[8:])
print(f"K-loop wave 0: {tk} cycles (s_memtime)")
for n_, v in zip(kn, s[:8]):
    if v:
        print(f"   {n_:38s} {v:9d}  {100.0 * v / tk:5.1f} %")
print(f"epilogue wave 4: {te} cycles")
for n_, v in zip(en, s[8:]):
    if v:
        print(f"   {n_:38s} {v:9d}  {100.0 * v / te:5.1f} %")
