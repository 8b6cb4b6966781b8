"""Which torch-side fills / zero-initialisations one bench step issues (each is a 3 us launch on the main stream).
Mirrors bench.py's default step: direct gradients, multi-task CE in one launch, FusedAdamW."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from linnaeus_amd.loss import multitask_cross_entropy
from linnaeus_amd.optim import FusedAdamW


class A:
    arch = "sm"; img = 224


cfg, model = bench.make_model(A)
model = model.cuda(); model.set_compute_dtype("bf16"); model.train(); model.grad_mode = "direct"
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
B = 32
x = torch.rand(B, 3, 224, 224, device="cuda"); meta = torch.rand(B, 5, device="cuda")
tg = {t: torch.randint(1, c, (B,), device="cuda") for t, c in bench.TASKS}


def step():
    model.zero_grad(set_to_none=True)
    out = model(x, meta)
    multitask_cross_entropy(out, tg).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::ones_like", "aten::empty_like", "aten::copy_", "aten::bernoulli_", "aten::mul", "aten::div_"):
        st = [f for f in (e.stack or []) if "linnaeus_amd" in f or "bench" in f]
        cnt[(e.name, str(e.input_shapes)[:50], (st[0][-60:] if st else ""))] += 1
for k, v in cnt.most_common(30):
    print(v, k)
