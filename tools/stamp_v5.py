"""Phase timeline of gemm_nt_v5 workgroups (diagnostic build tools/libv5_ablate.so): per workgroup start / end of K loop /
end of epilogue on the 100 MHz clock + HW_ID.  Prints, per stagger value, how the phases of the workgroups that share a CU
overlap.  Run: LNX_LIB_PATH=tools/libv5_ablate.so python tools/stamp_v5.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linnaeus_amd import _lib as L

M, N, K = 50944, 1536, 384
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
c2 = torch.empty_like(out)
a = L.GemmArgs()
a.dtype, a.M, a.N, a.K = L.BF16, M, N, K
a.A, a.lda, a.W, a.ldw = C.c_void_p(A.data_ptr()), K, C.c_void_p(W.data_ptr()), K
a.bias, a.act, a.c2, a.ldc2 = C.c_void_p(bias.data_ptr()), L.ACT_GELU, C.c_void_p(c2.data_ptr()), N
a.C, a.ldc = C.c_void_p(out.data_ptr()), N
grid = ((M + 255) // 256) * (N // 128)
stamps = torch.zeros(grid, 4, dtype=torch.int64, device="cuda")
os.environ["LNX_NT_V5"] = "1"
os.environ["LNX_V5_STAMPS"] = str(stamps.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for sg in sys.argv[1:] or ["0", "16"]:
    os.environ["LNX_V5_STAGGER"] = sg
    for _ in range(3):
        L.check(L.lib().lnx_gemm_nt(C.byref(a), st), "nt")
    torch.cuda.synchronize()
    s = stamps.cpu()
    t0 = s[:, 0].min()
    beg, mid, end, hw = (s[:, 0] - t0).float() / 100, (s[:, 1] - t0).float() / 100, (s[:, 2] - t0).float() / 100, s[:, 3]
    wave, simd, cu, sh, se, tg = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15
    print(f"stagger {sg}: {grid} workgroups, kernel {end.max():.1f} us; K loop {(mid - beg).mean():.2f} us mean, epilogue {(end - mid).mean():.2f} us mean")
    print("  wave slots used:", sorted(set(wave.tolist())), " tg ids:", sorted(set(tg.tolist()))[:16])
    # fraction of chip-time with k workgroups in the K loop / epilogue
    ev = torch.linspace(0, float(end.max()), 400)
    ink = ((beg[None, :] <= ev[:, None]) & (ev[:, None] < mid[None, :])).sum(1)
    ine = ((mid[None, :] <= ev[:, None]) & (ev[:, None] < end[None, :])).sum(1)
    for i in range(0, 400, 20):
        print(f"   t={ev[i]:6.1f} us: {int(ink[i]):4d} in K loop, {int(ine[i]):4d} in epilogue")
    first = beg < 1.0
    print(f"  first round: {int(first.sum())} workgroups start within 1 us; start spread of the rest: {beg[~first].min():.1f} .. {beg.max():.1f} us")
