"""One-GPU cost of running the backward segment by segment (what DataParallel does) against one call, at a given batch:
   python tools/bench_segments.py [batch] [--pg]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench as B

if "--pg" in sys.argv:  # does an initialised RCCL process group (world size 1, no collective ever issued) change anything?
    sys.argv.remove("--pg")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    lazy = "--lazy" in sys.argv
    if lazy:
        sys.argv.remove("--lazy")
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    print("process group initialised", "(lazily: no communicator yet)" if lazy else "(eagerly)")

class A: pass
args = A(); args.arch = "sm"; args.img = 224; args.batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg, model = B.make_model(args)
model = model.cuda(); model.set_compute_dtype("bf16"); model.train(); model.grad_mode = "direct"
from linnaeus_amd.optim import FusedAdamW
from linnaeus_amd.loss import multitask_cross_entropy
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(args.batch, 3, 224, 224, device="cuda", generator=g)
meta = torch.rand(args.batch, 5, device="cuda", generator=g)
tg = {t: torch.randint(1, c, (args.batch,), device="cuda", generator=g) for t, c in B.TASKS}
def step():
    model.zero_grad(set_to_none=True)
    loss = multitask_cross_entropy(model(x, meta), tg)
    loss.backward()
    opt.step()
def timeit(n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("one backward call      :", round(timeit(), 3), "ms/step")
if "dist" in globals() and dist.is_initialized():
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    print("after one all_reduce   :", round(timeit(), 3), "ms/step")
model._segment_hook = lambda s: None
print("four segments, no hook :", round(timeit(), 3), "ms/step")
ev = []
def hook(s):
    e = torch.cuda.Event(); e.record(torch.cuda.current_stream()); ev.append(e)
model._segment_hook = hook
print("four segments + event  :", round(timeit(), 3), "ms/step")
