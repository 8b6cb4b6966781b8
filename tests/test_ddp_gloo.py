"""N > 1 path on CPU: two gloo ranks drive the gradient-bucket reducer and the parameter
broadcast exactly as DataParallel does on the GPU (minus the HIP streams)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linnaeus_amd.ddp import DataParallel, GradBucketReducer, broadcast_module_state


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # flat arena with 4 segment buckets of uneven size (one empty)
        bounds = {0: (0, 40), 1: (40, 100), 2: (100, 100), 3: (100, 128)}
        arena = torch.arange(128, dtype=torch.float32) * (rank + 1)
        red = GradBucketReducer(arena, bounds)
        for seg in range(4):  # segments become ready in backward order
            red.reduce_bucket(seg)
        red.finish()
        expect = torch.arange(128, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok_avg = torch.allclose(arena, expect)
        # bf16-compressed buckets are a GPU-side option; on CPU the flag must not change results' shape
        # parameter/buffer broadcast from rank 0
        torch.manual_seed(100 + rank)
        m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        broadcast_module_state(m, 0)
        flat = torch.cat([t.detach().flatten().float() for t in list(m.parameters()) + list(m.buffers())])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok_bcast = all(torch.equal(gathered[0], g) for g in gathered)

        # DataParallel wiring with a stand-in module exposing the model's reducer interface
        class Fake(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.full((8,), float(rank)))
                self.grad_mode = "autograd"
                self._segment_hook = None
                self._grad_arena = torch.full((16,), float(rank + 1))
                self._segment_bounds = {0: (0, 4), 1: (4, 8), 2: (8, 12), 3: (12, 16)}

            def run_backward(self):
                for seg in range(4):
                    self._segment_hook(seg)

        fm = Fake()
        dp = DataParallel(fm)
        ok_wire = fm.grad_mode == "direct" and float(fm.w.detach()[0]) == 0.0  # broadcast from rank 0
        fm.run_backward()
        ok_wire &= torch.allclose(fm._grad_arena, torch.full((16,), (1 + world) / 2))
        with dp.no_sync():
            fm._grad_arena.fill_(float(rank))
            fm.run_backward()
        ok_nosync = torch.allclose(fm._grad_arena, torch.full((16,), float(rank)))
        q.put((rank, ok_avg, ok_bcast, ok_wire, ok_nosync))
    finally:
        dist.destroy_process_group()


def test_two_rank_bucket_allreduce_and_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    for rank, *flags in res:
        assert all(flags), (rank, flags)
