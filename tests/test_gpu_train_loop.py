"""All the pieces of one training loop together, the way the reference's train.py strings them (train.py:93-316,
h5data/h5dataloader.py:484-1341): host batches -> DevicePrefetcher -> GPU selective mixup -> drop-in mFormerV1 with
metadata tokens and hierarchical heads (recompute plan) -> weighted hierarchical loss with taxonomy-aware label smoothing
-> gradient accumulation -> FusedAdamW with clipping; then validation on an inference plan.  A small fixed data set has to
be memorised: the loss must fall and the accuracy rise.  Every piece has its own parity test; this one checks that they
compose (streams, plan cache, .grad views, accumulation, train/eval switches)."""
from types import SimpleNamespace as NS

import pytest
import torch

from linnaeus_amd import build_model
from oracle import mformer_oracle as O
from tests.cases import TINY_DIMS, TinyTree, make_config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("augment", [False, True])
def test_training_loop_pieces_compose_and_learn(augment):
    """augment=False: no mixup, no DropPath -- nothing stochastic is left but the summation order of the float
    atomics, and 32 separable images must be memorised outright (accuracy >= 0.95, loss below 0.3 of its start).
    augment=True: selective mixup + DropPath 0.1 on top; the bar there is that the loss falls and the accuracy rises."""
    from linnaeus_amd.collate import GPUSelectiveMixup
    from linnaeus_amd.loss import GradientWeighting, TaxonomyAwareLabelSmoothingCE, build_taxonomy_smoothing_matrix, weighted_hierarchical_loss
    from linnaeus_amd.optim import FusedAdamW
    from linnaeus_amd.prefetch import DevicePrefetcher

    import random

    random.seed(0)          # GPUSelectiveMixup flips its coin with Python's RNG
    torch.manual_seed(0)
    heads = (("taxa_L10", 6), ("taxa_L20", 3), ("taxa_L30", 2))
    tasks = [t for t, _ in heads]
    nc = dict(heads)
    spec = O.Spec(conv_dims=TINY_DIMS, conv_depths=(2, 1), rope_depths=(2, 1), rope_heads=(2, 4), meta=(("TEMPORAL", 2), ("SPATIAL", 3)),
                  heads=heads, drop_path_rate=0.1 if augment else 0.0)
    parent = {"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}}
    tree = TinyTree(parent, tasks, nc)
    cfg = make_config(spec, 64, "ConditionalClassifier")
    model = build_model(cfg, num_classes=nc, taxonomy_tree=tree).cuda()
    model.set_compute_dtype("bf16")
    model.use_checkpoint = True   # what train.py sets from TRAIN.GRADIENT_CHECKPOINTING (recompute plan)
    model.grad_mode = "direct"    # FusedAdamW loop: gradients accumulate in the flat arena

    # a fixed data set of 32 images whose label is readable from the image (mean brightness of a class-specific channel pattern)
    n, B = 32, 8
    gen = torch.Generator().manual_seed(1)
    y10 = torch.arange(n) % 6
    y20 = torch.tensor([parent["taxa_L10"][int(c)] for c in y10])
    y30 = torch.tensor([parent["taxa_L20"][int(c)] for c in y20])
    images = torch.rand(n, 3, 64, 64, generator=gen) * 0.2
    for i in range(n):
        c = int(y10[i])
        images[i, c % 3, (c // 3) * 32:(c // 3) * 32 + 32] += 0.8
    aux = torch.rand(n, 5, generator=gen)
    labels = {"taxa_L10": y10, "taxa_L20": y20, "taxa_L30": y30}

    def host_loader():
        for i in range(0, n, B):
            sl = slice(i, i + B)
            onehot = {t: torch.nn.functional.one_hot(labels[t][sl], nc[t]).float() for t in tasks}
            yield images[sl], onehot, aux[sl], torch.ones(B, 5, dtype=torch.bool), torch.zeros(B, dtype=torch.long)

    mix = GPUSelectiveMixup({"PROB": 0.5 if augment else 0.0, "ALPHA": 0.4, "meta_chunk_bounds_list": [(0, 2), (2, 5)]})
    crit = {}
    for t in tasks:
        dist = (1.0 - torch.eye(nc[t])).cuda()
        crit[t] = TaxonomyAwareLabelSmoothingCE(build_taxonomy_smoothing_matrix(nc[t], dist, alpha=0.05, beta=1.0)).cuda()
        crit[t].validate_targets = False
    lcfg = NS(TRAIN=NS(PHASE1_MASK_NULL_LOSS=False), LOSS=NS(GRAD_WEIGHTING=NS(CLASS=NS(TRAIN=False, VAL=False))))
    gw = GradientWeighting(tasks, lcfg, "static")
    sched = NS(get_null_mask_prob=lambda step: 1.0)
    opt = FusedAdamW(model.parameters(), lr=2e-3, weight_decay=0.01, max_grad_norm=1.0)
    accum = 2

    def accuracy():
        model.eval()
        hit = tot = 0
        with torch.no_grad():
            for img, tg, ax, *_ in DevicePrefetcher(host_loader()):
                out = model(img, ax)
                hit += int((out["taxa_L10"].argmax(-1) == tg["taxa_L10"].argmax(-1)).sum())
                tot += img.shape[0]
        model.train()
        return hit / tot

    model.train()
    acc0 = accuracy()
    epoch_loss = []
    step = 0
    for epoch in range(70):
        tot = 0.0
        opt.zero_grad(set_to_none=True)
        for bi, batch in enumerate(DevicePrefetcher(host_loader())):
            img, tg, ax, masks = mix(batch)
            out = model(img, ax)
            loss, comps, _ = weighted_hierarchical_loss(out, tg, crit, gw, sched, step, config=lcfg, sync_components=False)
            (loss / accum).backward()
            if (bi + 1) % accum == 0:
                opt.step()
                opt.zero_grad(set_to_none=True)
            tot += float(loss.detach())
            step += 1
        epoch_loss.append(tot / (n // B))
    acc1 = accuracy()
    print(f"[train loop] loss {epoch_loss[0]:.3f} -> {epoch_loss[-1]:.3f}; taxa_L10 accuracy {acc0:.2f} -> {acc1:.2f}; plans {len(model._plans)}")
    assert all(torch.isfinite(torch.tensor(epoch_loss)))
    if augment:
        assert epoch_loss[-1] < 0.6 * epoch_loss[0], epoch_loss
        assert acc1 >= 0.6 and acc1 > acc0 + 0.3, (acc0, acc1)  # mixup + DropPath draws amplify the summation-order noise of the float atomics
    else:
        assert epoch_loss[-1] < 0.3 * epoch_loss[0], epoch_loss
        assert acc1 >= 0.95, (acc0, acc1)
    # a recompute training plan and an inference plan, nothing else, are alive
    kinds = sorted((k[-2], k[-1]) for k in model._plans)
    assert kinds == [(False, False), (True, True)], kinds


def test_bench_line_loss_is_pinned_and_checked_against_the_oracle():
    """`bench.py` used to print a `loss` that nothing checked.  The driver's step counts (`--gpus 1 --steps 20 --warmup 5`;
    seeded inputs, seeded DropPath draws, 25 AdamW steps on one fixed batch; `--no-sched-calibration`, so that the count of untimed steps does
    not depend on how many schedules the warm-up compares) must reproduce the pinned loss of tests/golden/bench_loss.json within its stated
    tolerance (3x the run-to-run spread the float atomics of a few weight-gradient kernels leave after 25 steps) -- and the same process must
    have compared the timed model's loss with the CPU oracle's at the oracle's weights (`cpu_baseline.loss_check`; bench.py raises when
    they differ by more than its stated tolerance).  The line also carries config 3's per-GPU shape on one GPU (`config3_n1`)."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(repo, "tests", "golden", "bench_loss.json")) as fh:
        pin = json.load(fh)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-batch", "8", "--profile-steps", "0", "--no-sched-calibration"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"[bench line] loss {line['loss']} (pinned {pin['loss']}), loss_check {line['cpu_baseline']['loss_check']}, config3_n1 {line['config3_n1']['ms_per_step']} ms")
    assert abs(line["loss"] - pin["loss"]) <= pin["rel_tol"] * pin["loss"], (line["loss"], pin)
    chk = line["cpu_baseline"]["loss_check"]
    assert chk is not None and chk["abs_diff"] <= chk["tolerance"], chk
    n1 = line["config3_n1"]
    assert n1["per_gpu_batch"] == 128 and n1["ms_per_step"] > 0 and n1["images_per_sec"] > 0
    assert line["config"]["per_gpu_batch"] == 256 and line["roofline"] is None  # (--profile-steps 0)


def test_bench_n2_path_with_two_real_ranks_on_one_gpu():
    """`bench.py --gpus 2` end to end -- its own launcher (torch.distributed.run, one process per rank), barriers, max-over-ranks timing,
    DataParallel with the bucket all-reduces, the no_sync leg, the telemetry steps and the N > 1 fields of the line -- with two real ranks.
    RCCL refuses two ranks on one device, so `--rehearse-one-gpu` puts both on cuda:0 over gloo; the line says so and is not a measurement."""
    import json
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--batch", "8", "--steps", "2", "--warmup", "1",
                        "--profile-steps", "0", "--no-sched-calibration"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and "rehearsal" in line and line["scaling"] == "weak"
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["allreduce_of_ones"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["config"]["per_gpu_batch"] == 8 and line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert [b["bucket"] for b in line["rccl"]["buckets"]] == [0, 1, 2, 3] and all(b["bytes"] > 0 for b in line["rccl"]["buckets"])
    assert line["rccl"]["stream_budget"]["count"] <= 4
    dp = line["data_parallel"]
    assert dp["ms_per_step_no_sync"] > 0 and dp["n1_equiv_images_per_sec"] > 0 and line["value"] > 0
