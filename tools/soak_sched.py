"""Soak of the drawn-tile schedulers: N training steps of mFormerV1_sm at batch 256 (the shapes on which gemm_nt_v7 / v9 and the resident
conv-MLP kernels draw their tiles), same seed, once with LNX_TILE_SCHED=static and once with the atomic counters, each in its own process.
A tile processed twice or never would show as a loss trajectory that leaves the other one (the two agree to summation-order rounding);
a counter left non-zero would show as NaN / garbage from the next launch on.
`wgrad` as second argument: the same soak for the backward's weight-gradient stream (LNX_WGRAD_STREAM=0 against the default) -- a missing
join would let a weight-gradient product read a buffer its next writer already reached: wrong gradients from some step on.
usage: python tools/soak_sched.py [steps] [sched|wgrad] [batch]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import bench
from linnaeus_amd.loss import multitask_cross_entropy
from linnaeus_amd.optim import FusedAdamW
class A: arch = "sm"; img = 224
torch.manual_seed(0)
cfg, model = bench.make_model(A)
model = model.cuda(); model.set_compute_dtype("bf16"); model.train(); model.grad_mode = "direct"
opt = FusedAdamW(model.parameters(), lr=3e-4, weight_decay=0.05)
B = int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(B, 3, 224, 224, device="cuda", generator=g); meta = torch.rand(B, 5, device="cuda", generator=g)
tg = {t: torch.randint(1, c, (B,), device="cuda", generator=g) for t, c in bench.TASKS}
model._inject_drop = None
losses = []
for i in range(int(sys.argv[1])):
    torch.manual_seed(1000 + i)  # the DropPath draws of step i
    model.zero_grad(set_to_none=True)
    loss = multitask_cross_entropy(model(x, meta), tg)
    loss.backward()
    opt.step()
    if i %% 10 == 0 or i < 3:
        losses.append(loss.item())
print(json.dumps(losses))
''' % ROOT

steps = sys.argv[1] if len(sys.argv) > 1 else "150"
what = sys.argv[2] if len(sys.argv) > 2 else "sched"
batch = sys.argv[3] if len(sys.argv) > 3 else "256"
names = ("static", "atomic", "static2") if what == "sched" else ("serial", "stream", "serial2")
out = {}
for sched in names:
    base = sched.rstrip("2")
    env = dict(os.environ, LNX_TILE_SCHED=base) if what == "sched" else dict(os.environ, LNX_WGRAD_STREAM="0" if base == "serial" else "1")
    r = subprocess.run([sys.executable, "-c", CHILD, steps, batch], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        sys.exit(1)
    out[sched] = json.loads(r.stdout.strip().splitlines()[-1])
a, b, c = out[names[0]], out[names[1]], out[names[2]]
def spread(p_, q_):
    return [abs(u - v) / max(abs(u), 1e-9) for u, v in zip(p_, q_)]
rel, ctl = spread(a, b), spread(a, c)
for nm in names:
    print(f"{nm:8s}:", " ".join(f"{v:.4f}" for v in out[nm]))
print(f"max relative difference of the sampled losses: {names[1]} against {names[0]} {max(rel):.3e} (first step {rel[0]:.3e}); "
      f"CONTROL, {names[0]} against a second run of itself: {max(ctl):.3e} (first step {ctl[0]:.3e}); all finite: {all(v == v and abs(v) < 1e6 for v in a + b + c)}")
# Two runs of ONE schedule decorrelate once the loss is small (DESIGN 8b': atomics + Adam on a batch that is being memorised), so the later
# samples are judged against that control; the first steps are the sharp check (a schedule that computes something else differs at once).
early = max(rel[:3])
ok = all(v == v for v in a + b + c) and early < 1e-4 and max(rel) < max(0.15, 3.0 * max(ctl)) and b[-1] < 0.2 * b[0] and a[-1] < 0.2 * a[0]
sys.exit(0 if ok else 2)
