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


def _model_worker(rank, world, port, q):
    """The REAL mFormerV1 (construction and arena geometry are host-only): hierarchical heads that alias one Linear per task,
    two metadata components, its true bucket bounds from lnx_plan_segment_params."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from linnaeus_amd import build_model
        from tests.cases import CASES, TinyTree, make_config

        import dataclasses

        spec = dataclasses.replace(CASES["tiny_c"], meta=(("TEMPORAL", 2), ("SPATIAL", 3)))
        torch.manual_seed(1000 + rank)  # different initial weights per rank: the broadcast must equalise them
        tree = TinyTree({"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}},
                        [t for t, _ in spec.heads], {t: c for t, c in spec.heads})
        model = build_model(make_config(spec, 64, "ConditionalClassifier"), num_classes={t: c for t, c in spec.heads}, taxonomy_tree=tree)
        lay = model.grad_arena_layout(2)
        uniq = list(model.parameters())  # nn.Module.parameters() removes duplicates: shared head Linears once
        ok_unique = len({id(p_) for p_ in lay["params"]}) == len(lay["params"]) == len(uniq) and {id(p_) for p_ in uniq} == {id(p_) for p_ in lay["params"]}
        # every parameter owns one slice inside the bucket of its segment; slices are disjoint and 16-byte aligned
        spans = sorted((o, o + n, s) for o, n, s in zip(lay["offsets"], lay["numels"], lay["seg_of"]))
        ok_layout = all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and all(o % 4 == 0 for o, _, _ in spans)
        ok_layout &= all(lay["bounds"][s][0] <= o and e <= lay["bounds"][s][1] for o, e, s in spans)
        ok_layout &= [lay["bounds"][s] for s in range(4)] == sorted(lay["bounds"][s] for s in range(4)) and lay["bounds"][3][1] == lay["total"]
        # metadata heads are deferred one segment (side stream): stage-4 heads with bucket 1, stage-3 heads with bucket 2
        seg = dict(zip(lay["names"], lay["seg_of"]))
        ok_meta = all(v == (1 if "head_2" in k else 2) for k, v in seg.items() if k.startswith("meta."))
        ok_meta &= all(seg[k] == 0 for k in seg if k.startswith(("head.", "stages.3.", "final_norm", "norm_", "cl_1_fc", "aggregate")))
        ok_meta &= seg["stem.0.weight"] == 3 and seg["cls_token_1"] == 1 and any(k.startswith("meta.") for k in seg)

        arena = torch.zeros(lay["total"])
        for i, (o, n) in enumerate(zip(lay["offsets"], lay["numels"])):
            arena[o:o + n] = (rank + 1) * (i + 1)  # "gradient" of parameter i on this rank
        model._grad_arena, model._segment_bounds = arena, lay["bounds"]
        dp = DataParallel(model)  # broadcasts parameters and buffers from rank 0, sets direct mode, installs the segment hook
        flat = torch.cat([t.detach().flatten().float() for t in list(model.parameters()) + list(model.buffers())])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok_bcast = all(torch.equal(gathered[0], g) for g in gathered) and model.grad_mode == "direct"
        for s in range(4):  # what _plan_backward does after enqueueing each backward segment
            model._segment_hook(s)
        mean = sum(range(1, world + 1)) / world
        ok_avg = all(torch.allclose(arena[o:o + n], torch.full((n,), mean * (i + 1))) for i, (o, n) in enumerate(zip(lay["offsets"], lay["numels"])))
        with dp.no_sync():
            arena.fill_(float(rank))
            for s in range(4):
                model._segment_hook(s)
        ok_nosync = torch.equal(arena, torch.full_like(arena, float(rank)))
        q.put((rank, ok_unique, ok_layout, ok_meta, ok_bcast, ok_avg, ok_nosync))
    finally:
        dist.destroy_process_group()


def _run_workers(target, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    for rank, *flags in res:
        assert all(flags), (rank, flags)


def test_two_rank_real_model_bucket_geometry():
    """the real model's arena layout (`mFormerV1.grad_arena_layout`: `lnx_plan_segment_params` through the
    C ABI, no GPU) all-reduced by DataParallel across two gloo ranks."""
    _run_workers(_model_worker)


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
