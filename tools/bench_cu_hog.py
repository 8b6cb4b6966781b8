"""Step time of the data-parallel path beside a stand-in collective that holds CUs, on ONE GPU.

Where linnaeus_amd/ddp.py issues the all-reduce of a finished backward segment (reduce_bucket, on the communication stream behind
an event), tools/libcu_hog.so's kernel runs instead: `nwg` workgroups stream that bucket's bytes HBM -> HBM for the time a ring
all-reduce of the bucket would take over one xGMI link per direction (2 (n-1)/n bytes / 153 GB/s, n = 8), each claiming
--hog-lds bytes of LDS (so the 144-KiB-LDS persistent GEMM workgroups cannot be placed on its CUs).

    python tools/bench_cu_hog.py [--batch 128] [--wgs 0,32,64] [--steps 20]

Prints one line per hog size: ms/step, slowdown against wgs = 0 (no collective at all), and slowdown / (wgs / 256) -- 1.0 means the
step lost exactly the hog's share of the chip, 2+ means workgroups of the persistent kernels waited whole rounds for a free CU.
A/B: LNX_TILE_SCHED=static (round 3's `tile += gridDim.x`) against the default atomic tile counter; LNX_CU_MARGIN=n."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from linnaeus_amd.ddp import DataParallel  # noqa: E402
from linnaeus_amd.loss import multitask_cross_entropy  # noqa: E402
from linnaeus_amd.optim import FusedAdamW  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--wgs", default="0,32,64")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--hog-lds", type=int, default=64 * 1024)
ap.add_argument("--link-gbs", type=float, default=153.0)
ap.add_argument("--ranks", type=int, default=8)
a = ap.parse_args()

hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcu_hog.so"))
hog.hog_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_void_p]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
args = argparse.Namespace(arch="sm", img=224)
_, model = bench.make_model(args)
model = model.to(dev).train()
model.set_compute_dtype("bf16")
B = a.batch
g = torch.Generator(device=dev).manual_seed(42)
x = torch.rand(B, 3, 224, 224, device=dev, generator=g)
meta = torch.rand(B, 5, device=dev, generator=g)
tg = {t: torch.randint(1, c, (B,), device=dev, generator=g) for t, c in bench.TASKS}
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
scratch = {}
base = None
print(f"# batch {B}, LNX_TILE_SCHED={os.environ.get('LNX_TILE_SCHED', '(default)')}, LNX_CU_MARGIN={os.environ.get('LNX_CU_MARGIN', '(default)')}, hog LDS {a.hog_lds} B")
for nwg in (int(v) for v in a.wgs.split(",")):
    def collective(buf, nwg=nwg):
        nbytes = buf.numel() * 4
        us = 2.0 * (a.ranks - 1) / a.ranks * nbytes / (a.link_gbs * 1e9) * 1e6
        dst = scratch.get(buf.numel())
        if dst is None:
            dst = scratch[buf.numel()] = torch.empty_like(buf)
        rc = hog.hog_copy(buf.data_ptr(), dst.data_ptr(), nbytes // 16 * 16, nwg, us, a.hog_lds, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    net = DataParallel(model, broadcast=False, single_rank_collectives=nwg > 0, collective=collective if nwg > 0 else None)

    def step():
        model.zero_grad(set_to_none=True)
        multitask_cross_entropy(net(x, meta), tg).backward()
        opt.step()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    if nwg == 0:
        base = ms
        print(f"hog   0 workgroups: {ms:7.3f} ms/step")
    else:
        slow = ms / base - 1.0
        print(f"hog {nwg:3d} workgroups: {ms:7.3f} ms/step  slowdown {slow * 100:5.1f} %  = {slow / (nwg / 256):4.2f} x (hog CUs / 256)")
