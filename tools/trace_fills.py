"""Where do the small torch-side launches of a training step come from?  Runs a few bench-like steps under torch.profiler with Python stacks and
prints, per aten op that launches a fill / copy kernel, the innermost repo frames.  usage: python tools/trace_fills.py [--batch 128]"""
import argparse
import collections
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
sys.argv = ["bench.py"]
args = bench.parse()
torch.manual_seed(42)
cfg, model = bench.make_model(args)
model = model.cuda()
model.set_compute_dtype("bf16")
model.train()
model.grad_mode = "direct"
from linnaeus_amd.loss import multitask_cross_entropy  # noqa: E402
from linnaeus_amd.optim import FusedAdamW  # noqa: E402

opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
B = a.batch
x = torch.rand(B, 3, 224, 224, device="cuda")
meta = torch.rand(B, 5, device="cuda")
tg = {t: torch.randint(1, c, (B,), device="cuda") for t, c in bench.TASKS}


def step():
    model.zero_grad(set_to_none=True)
    loss = multitask_cross_entropy(model(x, meta), tg)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::ones", "aten::full", "aten::clone", "aten::bernoulli_", "aten::div_", "aten::to", "aten::_to_copy") and ev.device_type == torch.autograd.DeviceType.CPU:
        frames = [f for f in (ev.stack or []) if "/linnaeus_amd/" in f or "bench" in f or "trace_fills" in f]
        cnt[(ev.name, tuple(frames[:2]))] += 1
for (name, frames), n in cnt.most_common(25):
    print(f"{n / 2:6.1f}/step  {name:18s} {' <- '.join(f.split('/')[-1] for f in frames)}")
