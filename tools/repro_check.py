"""Run-to-run spread of the training step: two runs from the same seed (model init, inputs, DropPath draws), N steps each; prints the
relative difference of the loss trajectories and of the final parameters.  (DESIGN.md section 8b': the gradients are summed with
atomics in a few places, so runs agree to rounding, not bit for bit.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from linnaeus_amd.loss import multitask_cross_entropy
from linnaeus_amd.optim import FusedAdamW


class A:
    arch = "sm"; img = 224


def run(steps, B=32):
    torch.manual_seed(0)
    import random; random.seed(0)
    cfg, model = bench.make_model(A)
    model = model.cuda(); model.set_compute_dtype("bf16"); model.train(); model.grad_mode = "direct"
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, 3, 224, 224, device="cuda", generator=g); meta = torch.rand(B, 5, device="cuda", generator=g)
    tg = {t: torch.randint(1, c, (B,), device="cuda", generator=g) for t, c in bench.TASKS}
    losses = []
    for _ in range(steps):
        model.zero_grad(set_to_none=True)
        out = model(x, meta)
        loss = multitask_cross_entropy(out, tg)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
    return losses, flat


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
la, pa = run(steps)
lb, pb = run(steps)
rel = [abs(a - b) / max(abs(a), 1e-12) for a, b in zip(la, lb)]
print("first losses", la[:3], "last", la[-1])
print(f"loss trajectory: max relative difference over {steps} steps = {max(rel):.3e}; first step {rel[0]:.3e}")
print(f"parameters after {steps} steps: relative L2 difference = {((pa - pb).norm() / pa.norm()).item():.3e}, bit-equal fraction = {(pa == pb).float().mean().item():.4f}")
