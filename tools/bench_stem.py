"""Time lnx_stem_fwd at the sm / B = 256 shape (training form: patches, pre-norm copy and statistics written) and print the
achieved rate on its algorithmic bytes (image read + fp32 row + bf16 pre-norm row + bf16 patch matrix + statistics)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linnaeus_amd import ops

B, Cin, H, W, Cout = 256, 3, 224, 224, int(sys.argv[1]) if len(sys.argv) > 1 else 96
x = torch.randn(B, Cin, H, W, device="cuda")
w = (torch.randn(Cout, 64, device="cuda") / 7).to(torch.bfloat16)
bias, lw, lb = torch.randn(Cout, device="cuda"), torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
M = B * (H // 4) * (W // 4)
y = torch.empty(M, Cout, device="cuda")
pre = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
pat = torch.empty(M, 64, device="cuda", dtype=torch.bfloat16)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
def run(label, by, **kw):
    f = lambda: ops.stem_fwd(x, w, bias, lw, lb, y, **kw)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e-3
    print(f"stem_fwd Cout={Cout} {label}: {t*1e6:7.1f} us  {by/t/1e9:6.0f} GB/s of {by/1e6:.0f} MB")


run("training form (patches, pre-norm copy, statistics)", x.numel() * 4 + M * Cout * 6 + M * 64 * 2 + M * 8, patches=pat, pre=pre, mean=mean, rstd=rstd)
run("inference form (fp32 rows only)", x.numel() * 4 + M * Cout * 4)
