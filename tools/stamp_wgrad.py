"""Diagnostic: where a 32-row tile of convmlp_wgrad spends its cycles (s_memtime sums per wave).
Build:  hipcc ... -DCW_STAMP convmlp_wgrad.hip -> a second library (tools/build_stamp.sh), loaded via LNX_LIB_PATH."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
import torch
from linnaeus_amd import _lib as L

lib = L.lib()
lib.lnx_convmlp_wgrad_ws_floats.restype = C.c_int64
for Cc, M in ((96, 802816), (192, 200704)):
    bf = torch.bfloat16
    ln = torch.randn(M, Cc, device="cuda").to(bf); dz = torch.randn(M, Cc, device="cuda").to(bf)
    w1 = (torch.randn(4 * Cc, Cc, device="cuda") / Cc**0.5).to(bf); w2t = (torch.randn(4 * Cc, Cc, device="cuda") / (4 * Cc)**0.5).to(bf)
    b1 = torch.zeros(4 * Cc, device="cuda")
    dw1 = torch.zeros(4 * Cc, Cc, device="cuda"); db1 = torch.zeros(4 * Cc, device="cuda"); dw2 = torch.zeros(Cc, 4 * Cc, device="cuda"); db2 = torch.zeros(Cc, device="cuda")
    n = lib.lnx_convmlp_wgrad_ws_floats(Cc, M)
    ws = torch.zeros(n, device="cuda")
    a = L.ConvMlpWgradArgs()
    a.dtype, a.M, a.C = 1, M, Cc
    a.ln, a.dz, a.w1, a.w2t, a.b1 = ln.data_ptr(), dz.data_ptr(), w1.data_ptr(), w2t.data_ptr(), b1.data_ptr()
    a.dw1, a.db1, a.dw2, a.db2, a.ws, a.ws_floats = dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), n
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        L.check(lib.lnx_convmlp_wgrad(C.byref(a), st), "wgrad")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.check(lib.lnx_convmlp_wgrad(C.byref(a), st), "wgrad")
    e1.record(); torch.cuda.synchronize()
    print(f"C={Cc} M={M}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call (both matrices + reduces)")
    # the LAST launch was WHICH = 2; its slab 0 / split 0 head holds the stamps of 8 waves x 6 phases
    v = ws[:64].cpu().view(8, 8)[:, :6]
    names = ["commit", "barrier1", "phase1", "phase2", "(pre-b2)", "barrier2"]
    tot = v.sum(1)
    for w in range(8):
        print(f"  wave {w}: total {tot[w].item():10.0f} clk   " + "  ".join(f"{nm} {v[w, i].item() / tot[w].item() * 100:5.1f}%" for i, nm in enumerate(["commit", "barrier1", "phase1", "phase2", "barrier2", "-"]) if i < 5))
