"""Debug helper: fused conv-MLP weight gradients vs fp64 for one (C, M); prints where the errors are."""
import sys
import torch
sys.path.insert(0, ".")
from linnaeus_amd import ops

def run(C_, M):
    gen = torch.Generator().manual_seed(C_ * 3 + M)
    bf = torch.bfloat16
    ln = torch.randn(M, C_, generator=gen).cuda().to(bf)
    dz = torch.randn(M, C_, generator=gen).cuda().to(bf)
    w1 = (torch.randn(4 * C_, C_, generator=gen) / C_**0.5).cuda().to(bf)
    b1 = (0.2 * torch.randn(4 * C_, generator=gen)).cuda()
    w2 = (torch.randn(C_, 4 * C_, generator=gen) / (4 * C_) ** 0.5).cuda().to(bf)
    dw1 = torch.zeros(4 * C_, C_, device="cuda"); db1 = torch.zeros(4 * C_, device="cuda")
    dw2 = torch.zeros(C_, 4 * C_, device="cuda"); db2 = torch.zeros(C_, device="cuda")
    ops.convmlp_wgrad(ln, dz, w1, w2.t().contiguous(), b1, dw1, db1, dw2, db2)
    torch.cuda.synchronize()
    h = (ln.double() @ w1.double().T + b1.double()).requires_grad_(True)
    act = torch.nn.functional.gelu(h)
    act.sum().backward()
    act64 = act.detach().to(bf).double()
    dh64 = ((dz.double() @ w2.double()) * h.grad).to(bf).double()
    r1 = dh64.T @ ln.double(); r2 = dz.double().T @ act64
    e1 = (dw1.double() - r1).abs(); e2 = (dw2.double() - r2).abs()
    print(f"C={C_} M={M}: dW1 maxerr {e1.max().item():.3e} (ref max {r1.abs().max().item():.2f}) dW2 maxerr {e2.max().item():.3e} "
          f"db1 {(db1.double() - dh64.sum(0)).abs().max().item():.3e} db2 {(db2.double() - dz.double().sum(0)).abs().max().item():.3e}")
    bad1 = (e1 > 0.05 * r1.abs().max()).nonzero()
    bad2 = (e2 > 0.05 * r2.abs().max()).nonzero()
    if len(bad1):
        print("  dW1 bad count", len(bad1), "rows(j)", sorted(set(bad1[:, 0].tolist()))[:40], "cols(c)", sorted(set(bad1[:, 1].tolist()))[:40])
    if len(bad2):
        print("  dW2 bad count", len(bad2), "rows(c)", sorted(set(bad2[:, 0].tolist()))[:40], "cols(j)", sorted(set(bad2[:, 1].tolist()))[:40])

for C_, M in [(32, 32), (32, 200), (64, 130), (96, 777), (128, 100), (192, 333)]:
    run(C_, M)
