"""End-to-end parity of the HIP mFormerV1 (through build_model() and the C-ABI plan) against
the golden fixtures produced by the reference, and against the CPU oracle's gradients.

Tolerances (stated per BASELINE.md section 4):
  fp32 mode : logits rtol 1e-4 / atol 5e-5 vs the reference, class-index argmax EXACT
  bf16 mode : max-abs logit error reported and bounded; argmax exact wherever the
              reference's top-1/top-2 margin exceeds 4x the observed error
"""
import numpy as np
import pytest
import os

import torch

from linnaeus_amd import build_model
from oracle import mformer_oracle as O
from tests.cases import CASES, TinyTree, load_case, make_config, model_state_dict_from_oracle

pytestmark = pytest.mark.gpu
IMG = {"tiny_a": 64, "tiny_b": 96, "tiny_c": 64, "tiny_dp": 64, "sm": 224}


def build(name, spec, sd, dtype):
    kw = {"num_classes": {t: c for t, c in spec.heads}}
    head_type = "Linear"
    if name == "tiny_c":
        head_type = "ConditionalClassifier"
        kw["taxonomy_tree"] = TinyTree({"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}},
                                       [t for t, _ in spec.heads], {t: c for t, c in spec.heads})
    model = build_model(make_config(spec, IMG[name], head_type), **kw)
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    return model


def run(model, x, meta, drops, train):
    model.train(train)
    model._inject_drop = drops
    return model(x.cuda(), meta.cuda() if meta is not None else None)


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_dp", "sm"])
def test_forward_fp32_matches_reference(name, golden_dir):
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    model = build(name, spec, sd, "fp32")
    with torch.no_grad():
        out = run(model, x, meta, drops, train=drops is not None)
        feats = model._last_feats
    np.testing.assert_allclose(feats.cpu().numpy(), z["feats"], rtol=1e-4, atol=5e-5)
    for task, _ in spec.heads:
        ref = z["logits_" + task]
        got = out[task].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=5e-5, err_msg=task)
        assert (got.argmax(-1) == ref.argmax(-1)).all(), f"argmax differs for {task}"


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_dp", "sm"])
def test_forward_bf16_close_to_reference(name, golden_dir):
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    model = build(name, spec, sd, "bf16")
    with torch.no_grad():
        out = run(model, x, meta, drops, train=drops is not None)
    worst = 0.0
    for task, _ in spec.heads:
        ref = z["logits_" + task]
        got = out[task].float().cpu().numpy()
        err = np.abs(got - ref).max()
        worst = max(worst, err)
        scale = np.abs(ref).max()
        assert err <= 0.025 * max(scale, 1.0), f"{task}: bf16 max-abs error {err:.4f} vs logit scale {scale:.3f}"  # 3x the largest measured (0.008 of scale)
        srt = np.sort(ref, -1)
        margin = srt[:, -1] - srt[:, -2]
        safe = margin > 4 * err
        assert (got.argmax(-1)[safe] == ref.argmax(-1)[safe]).all()
    print(f"[{name}] bf16 max-abs logit error vs reference fp32: {worst:.5f}")


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_dp"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_backward_matches_oracle(name, dtype, golden_dir):
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    model = build(name, spec, sd, dtype)
    out = run(model, x, meta, drops, train=True)
    loss = O.probe_loss({k: v for k, v in out.items()})
    if dtype == "fp32":
        assert abs(loss.item() - float(z["loss"])) < 2e-4 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    # oracle gradients on the CPU
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.probe_loss(O.forward(osd, spec, x, meta, drops)).backward()
    names = [str(n) for n in z["grad_names"]]
    got = {}
    for k, p_ in model.named_parameters():
        parts = k.split(".")
        ck = f"head.{parts[3]}.fc.{parts[4]}" if (parts[0] == "head" and len(parts) >= 5 and parts[2] == "level_classifiers") else k
        got[ck] = p_.grad
    assert sorted(got) == names
    # fp32: tight per-parameter bound.  bf16: operands/activations are rounded to 8 mantissa bits; per parameter 10 % (3x the largest
    # measured outside the exception), the whole gradient 5 %.  The exception, by name: the metadata-head chains (meta_*_head_*:
    # Linear -> ReLU -> LN -> ResNorm on a batch of 2-4 rows, mFormerV1.py:282-311) -- one pre-activation that rounds across zero flips
    # a ReLU mask and with it a whole row's contribution: 25 %.
    def rel_tol_of(name):
        if dtype == "fp32":
            return 2e-3
        return 0.25 if name.startswith("meta_") else 0.10
    bad = []
    worst = []
    tot_err = tot_ref = 0.0
    for i, k in enumerate(names):
        g = got[k].float().cpu()
        ref = osd[k].grad
        denom = ref.norm().item()
        err = (g - ref).norm().item()
        tot_err += err * err
        tot_ref += denom * denom
        if dtype == "fp32":  # the fixture pins the oracle; check the GPU against the reference's own numbers too
            assert abs(g.double().norm().item() - z["grad_norms"][i]) <= 5e-3 * max(z["grad_norms"][i], 1e-3), k
        worst.append((err / max(denom, 1e-3 * (1 if dtype == "fp32" else 10)), k))
        if err > rel_tol_of(k) * max(denom, 1e-3 * (1 if dtype == "fp32" else 10)):
            bad.append((k, err, denom))
    worst.sort(reverse=True)
    print(f"[{name}/{dtype}] worst per-parameter relative gradient errors: " + ", ".join(f"{k} {e:.3f}" for e, k in worst[:4]))
    assert not bad, bad[:8]
    glob = (tot_err / tot_ref) ** 0.5
    print(f"[{name}/{dtype}] global relative gradient error vs oracle: {glob:.2e}")
    assert glob <= (1e-3 if dtype == "fp32" else 5e-2)


def test_sm_backward_grad_norms_fp32(golden_dir):
    spec, z, sd, x, meta, drops = load_case("sm", golden_dir)
    model = build("sm", spec, sd, "fp32")
    out = run(model, x, meta, None, train=True)
    O.probe_loss(out).backward()
    names = [str(n) for n in z["grad_names"]]
    got = dict(model.named_parameters())
    for i, k in enumerate(names):
        n = got[k].grad.double().norm().item()
        assert abs(n - z["grad_norms"][i]) <= 5e-3 * max(z["grad_norms"][i], 1e-3), (k, n, z["grad_norms"][i])
        np.testing.assert_allclose(got[k].grad.reshape(-1)[:8].cpu().numpy(), z["gradslice_" + k], rtol=2e-2, atol=2e-5, err_msg=k)


def test_state_dict_contract_and_errors():
    spec = CASES["tiny_a"]
    model = build_model(make_config(spec, 64), num_classes={t: c for t, c in spec.heads})
    shapes = O.param_shapes(spec)
    sd = model.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    with pytest.raises(Exception, match="no CPU path|GPU"):
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, 5))
    model = model.cuda()
    with pytest.raises(AssertionError):
        model(torch.zeros(1, 3, 64, 64, device="cuda"), None)  # meta configured but missing


def test_direct_grad_mode_accumulates():
    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 1)
    model = build("tiny_a", spec, sd, "fp32")
    model.grad_mode = "direct"
    x, meta = O.seeded_inputs(spec, 2, 64, 5)
    model.train()
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()
    g1 = {k: p_.grad.clone() for k, p_ in model.named_parameters()}
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()  # accumulate (no zero_grad)
    for k, p_ in model.named_parameters():
        torch.testing.assert_close(p_.grad, 2 * g1[k], rtol=1e-4, atol=1e-6)
    model.zero_grad(set_to_none=True)
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()
    for k, p_ in model.named_parameters():
        torch.testing.assert_close(p_.grad, g1[k], rtol=1e-4, atol=1e-6)


def test_torch_ddp_wrap_default_grad_mode():
    """an untouched model wrapped in torch DistributedDataParallel -- what the reference's launch path does
    (main.py:982).  The default grad_mode returns gradients through autograd, so DDP's AccumulateGrad hooks fire, the reduction
    finishes and a second forward is accepted; gradients equal the un-wrapped model's.  One rank over RCCL."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 1)
    x, meta = O.seeded_inputs(spec, 2, 64, 5)
    plain = build("tiny_a", spec, sd, "fp32")
    assert plain.grad_mode == "autograd"
    plain.train()
    O.probe_loss(plain(x.cuda(), meta.cuda())).backward()
    want = {k: p_.grad.clone() for k, p_ in plain.named_parameters()}
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        model = build("tiny_a", spec, sd, "fp32")
        model.train()
        ddp = DDP(model, device_ids=[torch.cuda.current_device()], find_unused_parameters=True)  # as main.py:975-982
        fired = []
        next(model.parameters()).register_post_accumulate_grad_hook(lambda p_: fired.append(1))
        for step in range(2):  # the second forward raises "Expected to have finished reduction" if no hook fired in the first
            ddp.zero_grad(set_to_none=True)
            O.probe_loss(ddp(x.cuda(), meta.cuda())).backward()
            for k, p_ in model.named_parameters():
                torch.testing.assert_close(p_.grad, want[k], rtol=1e-4, atol=1e-6, msg=f"step {step} {k}")
        assert len(fired) == 2
        g = torch.autograd.grad(O.probe_loss(model(x.cuda(), meta.cuda())), [model.cls_token_2])[0]  # GradNorm-style caller
        torch.testing.assert_close(g, want["cls_token_2"], rtol=1e-4, atol=1e-6)
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_train_step_matches_reference(dtype, golden_dir):
    """Caller (ii) of SURVEY 8c on the HIP model: forward -> CE -> backward -> clip_grad_norm_ -> AdamW, two
    steps, against the numbers the reference produced (tests/golden/train_step.npz)."""
    from tests.cases import load_train_step, train_steps

    spec, z, sd, x, meta, targets, weights = load_train_step(golden_dir)
    model = build("tiny_a", spec, sd, dtype)
    model.train()
    xg, mg = x.cuda(), meta.cuda()
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    losses, norms = train_steps(lambda: model(xg, mg), list(model.parameters()), z, spec, targets, weights)
    ltol, ntol, dtol = (2e-4, 2e-3, 5e-2) if dtype == "fp32" else (2e-2, 5e-2, 0.3)
    for s in range(int(z["steps"])):
        assert abs(losses[s] - float(z[f"loss_{s}"])) <= ltol * abs(float(z[f"loss_{s}"])), (s, losses[s], float(z[f"loss_{s}"]))
        assert abs(norms[s] - float(z[f"gnorm_{s}"])) <= ntol * float(z[f"gnorm_{s}"]), (s, norms[s], float(z[f"gnorm_{s}"]))
    names = [str(n) for n in z["param_names"]]
    got = dict(model.named_parameters())
    assert sorted(got) == names
    # Adam's first steps move every element by ~lr*sign(g): delta norms are insensitive to gradient scale but flip with
    # gradient sign, so they pin the direction of every update
    for i, k in enumerate(names):
        d = (got[k].detach() - before[k]).double().norm().item()
        if k == "aggregate.bias":
            continue  # a constant shift in front of final_norm: its gradient is exactly zero in real arithmetic, and Adam turns the rounding noise into +-lr
        floor = 0.05 * float(z["lr"]) * int(z["steps"]) * got[k].numel() ** 0.5
        assert abs(d - z["delta_norms"][i]) <= dtol * z["delta_norms"][i] + floor, (k, d, z["delta_norms"][i])
        if dtype == "fp32" and z["delta_norms"][i] > 0.5 * float(z["lr"]) * got[k].numel() ** 0.5:
            ref = z["deltaslice_" + k]
            cur = (got[k].detach() - before[k]).reshape(-1)[: ref.size].cpu().numpy()
            assert np.abs(cur - ref).max() <= 0.2 * np.abs(ref).max() + 1e-7, (k, cur, ref)


def test_classification_heads_in_several_groups(golden_dir, monkeypatch):
    """The plan issues every classification head in one grouped launch, forward and data gradient (lnx_gemm_nt_group); a model with more
    heads than a launch carries (8) goes in several groups, the data-gradient groups accumulating onto each other.  LNX_HEADS_GROUP_MAX
    (read per call) makes the groups smaller, so the sm fixture's four heads walk that path: groups of 3 + 1 and of 1 + 1 + 1 + 1 give
    the same logits bit for bit (independent products) and the same gradients up to the rounding of the accumulator chain."""
    spec, z, sd, x, meta, drops = load_case("sm", golden_dir)
    model = build("sm", spec, sd, "bf16")

    def step():
        model.zero_grad(set_to_none=True)
        out = run(model, x, meta, drops, train=True)
        sum((out[t].float() ** 2).mean() for t, _ in spec.heads).backward()
        return {t: out[t].detach().clone() for t, _ in spec.heads}, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    want_out, want_g = step()
    for gmax in ("3", "1"):
        monkeypatch.setenv("LNX_HEADS_GROUP_MAX", gmax)
        got_out, got_g = step()
        for t, _ in spec.heads:
            assert torch.equal(got_out[t], want_out[t]), (gmax, t)
        for k, g_ in want_g.items():
            if k == "aggregate.bias":
                continue  # (exactly zero in real arithmetic -- a constant shift in front of final_norm: what is compared would be rounding noise)
            torch.testing.assert_close(got_g[k], g_, rtol=2e-4, atol=2e-4 * float(g_.abs().max()) + 1e-12, msg=lambda m, k=k: f"{k} (groups of {gmax}): {m}")


def test_large_384_matches_oracle():
    """BASELINE config 4's shape: mFormerV1_lg (dims 192..1536, rope depths 10/2, heads 12/24) at 384x384, so
    N = 24*24 + 3 = 579 tokens in stage 2 (multi-tile attention path) and conv stages the fused C<=192 kernels only
    partly cover.  Forward (fp32 storage) and gradients against the CPU oracle on the same seeded weights."""
    spec = O.Spec(conv_dims=(192, 384, 768, 1536), rope_depths=(10, 2), rope_heads=(12, 24), heads=(("taxa_L10", 50), ("taxa_L20", 11)))
    sd = O.seeded_state_dict(O.param_shapes(spec), 1234)
    x, meta = O.seeded_inputs(spec, 1, 384, 99)
    model = build_model(make_config(spec, 384), num_classes={t: c for t, c in spec.heads})
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for v in sd.values())
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta)
    O.probe_loss(oout).backward()
    # bf16 bounds = 3x the measured error (0.005 of the logit scale, 0.0084 of the gradient norm; both printed), not a loose 0.10
    for dtype, ftol, gtol in (("fp32", 2e-4, 2e-3), ("bf16", 0.015, 0.025)):
        model.set_compute_dtype(dtype)
        model.train()
        model.zero_grad()
        out = model(x.cuda(), meta.cuda())
        for t, _ in spec.heads:
            ref = oout[t].detach()
            err = (out[t].float().cpu() - ref).abs().max().item()
            print(f"[lg@384/{dtype}] {t}: max logit error / scale {err / max(1.0, ref.abs().max().item()):.4f}")
            assert err <= ftol * max(1.0, ref.abs().max().item()), (dtype, t, err)
            if dtype == "fp32":
                assert (out[t].argmax(-1).cpu() == ref.argmax(-1)).all()
        O.probe_loss(out).backward()
        tot_err = tot_ref = 0.0
        for k, p_ in model.named_parameters():
            ref = osd[k].grad
            tot_err += (p_.grad.float().cpu() - ref).double().pow(2).sum().item()
            tot_ref += ref.double().pow(2).sum().item()
        glob = (tot_err / tot_ref) ** 0.5
        print(f"[lg@384/{dtype}] global relative gradient error vs oracle: {glob:.2e}")
        assert glob <= gtol, (dtype, glob)


def test_xlarge_224_matches_oracle_and_autobatch():
    """BASELINE config 5: mFormerV1_xl (dims 256..2048, rope depths 22/2, heads 16/32) in fp32, bf16 and fp8 mode (MXFP8
    forward products in the 22 stage-3 RoPE blocks: 2 x 200 = 400 rows >= the fp8 kernel's 256-row floor; the two stage-4
    blocks have 106 rows and stay bf16) -- conv stages on the plain GEMM path (C = 256 / 512 is beyond the fused conv-MLP),
    LayerNorm at C = 1024 / 2048, 24 RoPE blocks.  Logits and gradients against the CPU oracle on the same seeded
    weights, then AutoBatch (utils/autobatch.py:111-265) sizes the batch for a memory budget from the planner's exact
    workspace figure.  Stated tolerances, as fractions of the logit scale / of the global gradient norm: bf16 0.04 / 0.05 (3x the
    measured 0.012 / 0.016), fp8 0.10 / 0.12 (measured 0.068 / 0.085 -- 22 quantised blocks deep on random-init weights)."""
    from linnaeus_amd.autobatch import auto_find_batch_size, foreign_bytes, predicted_bytes

    spec = O.Spec(conv_dims=(256, 512, 1024, 2048), rope_depths=(22, 2), rope_heads=(16, 32), heads=(("taxa_L10", 40), ("taxa_L20", 9)))
    sd = O.seeded_state_dict(O.param_shapes(spec), 4321)
    x, meta = O.seeded_inputs(spec, 2, 224, 77)
    cfg = make_config(spec, 224)
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for v in sd.values())
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta)
    O.probe_loss(oout).backward()
    for dtype, ftol, gtol in (("fp32", 3e-4, 3e-3), ("bf16", 0.04, 0.05), ("fp8", 0.10, 0.12)):
        model.set_compute_dtype(dtype)
        model.train()
        model.zero_grad()
        out = model(x.cuda(), meta.cuda())
        for t, _ in spec.heads:
            ref = oout[t].detach()
            err = (out[t].float().cpu() - ref).abs().max().item()
            print(f"[xl@224/{dtype}] {t}: max logit error / scale {err / max(1.0, ref.abs().max().item()):.4f}")
            assert err <= ftol * max(1.0, ref.abs().max().item()), (dtype, t, err)
            if dtype == "fp32":
                assert (out[t].argmax(-1).cpu() == ref.argmax(-1)).all()
        O.probe_loss(out).backward()
        tot_err = tot_ref = 0.0
        for k, p_ in model.named_parameters():
            ref = osd[k].grad
            tot_err += (p_.grad.float().cpu() - ref).double().pow(2).sum().item()
            tot_ref += ref.double().pow(2).sum().item()
        glob = (tot_err / tot_ref) ** 0.5
        print(f"[xl@224/{dtype}] global relative gradient error vs oracle: {glob:.2e}")
        assert glob <= gtol, (dtype, glob)
    # AutoBatch: analytic footprint is monotone, and the search returns the largest batch under a (small, so that the
    # verification step is quick) budget; at 0.9 of the 288 GB the same formula gives the production batch
    model.set_compute_dtype("bf16")
    model.zero_grad(set_to_none=True)
    total = torch.cuda.get_device_properties(0).total_memory
    other = foreign_bytes(model)
    b8, b9 = predicted_bytes(model, 8, 224), predicted_bytes(model, 9, 224)
    assert b9 > b8
    frac = (b8 + 0.5 * (b9 - b8) + other) / total
    assert auto_find_batch_size(model, cfg, "train", target_memory_fraction=frac, max_batch_size=512, steps_per_trial=1) == 8
    big = 8 + int((0.9 * total - other - b8) // (b9 - b8))
    print(f"[xl@224] AutoBatch at 0.9 x {total / 2**30:.0f} GiB: batch {big} per GPU ({(b9 - b8) / 2**20:.0f} MiB per image)")
    assert big > 256


def test_data_parallel_stream_logic_single_rank(golden_dir):
    """DataParallel on one GPU with the collectives forced on (RCCL, world size 1): the per-segment all-reduces on the
    side stream, their events and the final join must leave exactly the gradients of the plain model, for the plain
    and the bf16-compressed buckets, and with no_sync() accumulation."""
    import os
    import torch.distributed as dist
    from linnaeus_amd.ddp import DataParallel

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        spec, z, sd, x, meta, drops = load_case("tiny_b", golden_dir)
        ref = build("tiny_b", spec, sd, "fp32")
        ref.train()
        O.probe_loss(run(ref, x, meta, None, train=True)).backward()
        want = {k: p_.grad.clone() for k, p_ in ref.named_parameters()}
        for compress in (False, True):
            model = build("tiny_b", spec, sd, "fp32")
            model.train()
            dp = DataParallel(model, compress_bf16=compress, single_rank_collectives=True)
            for rep in range(2):  # second pass: gradients re-zeroed, same answer
                model.zero_grad(set_to_none=True)
                O.probe_loss(dp(x.cuda(), meta.cuda())).backward()
                torch.cuda.synchronize()
                for k, p_ in model.named_parameters():
                    tol = dict(rtol=1e-5, atol=1e-7) if not compress else dict(rtol=2e-2, atol=1e-4)
                    torch.testing.assert_close(p_.grad, want[k], msg=f"{k} (compress={compress}, pass {rep})", **tol)
            model.zero_grad(set_to_none=True)
            with dp.no_sync():
                O.probe_loss(dp(x.cuda(), meta.cuda())).backward()
            O.probe_loss(dp(x.cuda(), meta.cuda())).backward()
            torch.cuda.synchronize()
            if not compress:
                for k, p_ in model.named_parameters():
                    torch.testing.assert_close(p_.grad, 2 * want[k], rtol=1e-5, atol=1e-7, msg=f"{k} (accumulated)")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_odd_batches_plan_reuse_and_eval(dtype, golden_dir):
    """Batch sizes that leave every tile ragged (1, 7), alternating on ONE model object (a plan per batch size, shared
    parameters and gradient arena), eval-mode forwards in between, non-contiguous / half-precision inputs: logits and
    gradients against the oracle each time."""
    spec, z, sd, _, _, _ = load_case("tiny_a", golden_dir)
    model = build("tiny_a", spec, sd, dtype)
    ftol, gtol = (2e-4, 2e-3) if dtype == "fp32" else (0.08, 0.08)
    for B in (1, 7, 1, 2):
        x, meta = O.seeded_inputs(spec, B, 64, 1000 + B)
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        oout = O.forward(osd, spec, x, meta)
        O.probe_loss(oout).backward()
        # eval forward first (no autograd), from a channels-last view of the same values
        model.eval()
        with torch.no_grad():
            xe = x.cuda().to(memory_format=torch.channels_last)
            out_e = model(xe, meta.cuda())
        model.train()
        model.zero_grad(set_to_none=True)
        out = model(x.cuda(), meta.cuda())
        for t, _ in spec.heads:
            ref = oout[t].detach()
            for got in (out[t], out_e[t]):
                err = (got.float().cpu() - ref).abs().max().item()
                assert err <= ftol * max(1.0, ref.abs().max().item()), (B, t, err)
        O.probe_loss(out).backward()
        tot_err = tot_ref = 0.0
        for k, p_ in model.named_parameters():
            ref = osd[k].grad
            tot_err += (p_.grad.float().cpu() - ref).double().pow(2).sum().item()
            tot_ref += ref.double().pow(2).sum().item()
        assert (tot_err / tot_ref) ** 0.5 <= gtol, (B, (tot_err / tot_ref) ** 0.5)


# ----------------------------------------------------------------------------------------------------
# Round-2 additions: the TIMED configuration's kernel dispatch under the oracle
# ----------------------------------------------------------------------------------------------------
def _drop_scales(spec, B, seed):
    """Per-call, per-sample DropPath multipliers floor(keep + U)/keep (blocks/drop_path.py:29-33), None where p == 0."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for p_ in O.drop_call_probs(spec):
        if p_ == 0.0:
            out.append(None)
        else:
            keep = 1.0 - p_
            out.append(torch.floor(keep + torch.rand(B, generator=g)) / keep)
    return out


def _grad_errors(model, osd):
    tot_err = tot_ref = 0.0
    worst = ("", 0.0)
    for k, p_ in model.named_parameters():
        parts = k.split(".")
        ck = f"head.{parts[3]}.fc.{parts[4]}" if (parts[0] == "head" and len(parts) >= 5 and parts[2] == "level_classifiers") else k
        ref = osd[ck].grad
        e = (p_.grad.float().cpu() - ref).double().pow(2).sum().item()
        r = ref.double().pow(2).sum().item()
        tot_err += e
        tot_ref += r
        rel = (e / max(r, 1e-30)) ** 0.5
        if r > 1e-12 and rel > worst[1]:
            worst = (k, rel)
    return (tot_err / tot_ref) ** 0.5, worst


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sm_b24_production_dispatch_matches_oracle(dtype):
    """mFormerV1_sm @224, B = 24 with DropPath multipliers injected: RoPE M = 4,776 / 1,248 >= 1024 and conv M = 75,264 / 18,816,
    so the pipelined gemm_nt_v2/v4, gemm_tn_v2 + reduce, the resident C=96 / streamed C=192 fused conv-MLP kernels, the
    fused weight-gradient kernels and the side stream all run INSIDE the plan exactly as in bench.py's B = 256 step
    (same dispatch thresholds), against O.forward + probe_loss on the CPU.  Tolerances as stated at the top of this file."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20)), drop_path_rate=0.2)
    B = 24
    sd = O.seeded_state_dict(O.param_shapes(spec), 777)
    x, meta = O.seeded_inputs(spec, B, 224, 778)
    drops = _drop_scales(spec, B, 779)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta, drops)
    O.probe_loss(oout).backward()
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    out = run(model, x, meta, drops, train=True)
    worst = 0.0
    for t, _ in spec.heads:
        ref = oout[t].detach()
        got = out[t].float().cpu()
        err = (got - ref).abs().max().item()
        worst = max(worst, err)
        scale = max(1.0, ref.abs().max().item())
        if dtype == "fp32":
            torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * scale, msg=t)
            assert (got.argmax(-1) == ref.argmax(-1)).all(), t
        else:
            assert err <= 0.025 * scale, (t, err, scale)
            srt = ref.sort(-1).values
            safe = (srt[:, -1] - srt[:, -2]) > 4 * err
            assert (got.argmax(-1)[safe] == ref.argmax(-1)[safe]).all(), t
    O.probe_loss(out).backward()
    glob, wk = _grad_errors(model, osd)
    print(f"[sm B=24/{dtype}] max-abs logit error {worst:.5f}; global relative gradient error {glob:.2e}; worst tensor {wk[0]} {wk[1]:.2e}")
    assert glob <= (1e-3 if dtype == "fp32" else 5e-2), (glob, wk)
    if dtype == "fp32":
        assert wk[1] <= 5e-3, wk


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sm_b256_batch_invariance(dtype):
    """BASELINE config 2 at ITS OWN batch: mFormerV1_sm @224, B = 256, DropPath 0.2, the kernels bench.py times (in bf16 the
    persistent gemm_nt_v7 / v9 on the 597- / 1791-tile grids of M = 50 944, the 128x256 weight-gradient orientation and split-K
    counts of that M) -- checked through batch invariance: rows 0-23 of the batch are the oracle-checked B = 24 inputs (same
    DropPath multipliers), rows 24-255 seeded filler.  Every sample is independent in the reference (mFormerV1.py:407-541: no
    batch statistics anywhere), so (a) logits of rows 0-23 must equal the oracle's B = 24 logits within this file's bounds and the
    GPU's own B = 24 run to rounding, and (b) with the probe loss restricted to rows 0-23 every parameter gradient must equal the
    B = 24 gradient (rope_2d_mhsa.py:422-505, blocks/mlp.py:46-66 and their autograd backward)."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20)), drop_path_rate=0.2)
    B0, B = 24, 256
    sd = O.seeded_state_dict(O.param_shapes(spec), 777)
    x0, meta0 = O.seeded_inputs(spec, B0, 224, 778)
    drops0 = _drop_scales(spec, B0, 779)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x0, meta0, drops0)
    O.probe_loss(oout).backward()
    xf, metaf = O.seeded_inputs(spec, B - B0, 224, 780)
    dropsf = _drop_scales(spec, B - B0, 781)
    x, meta = torch.cat([x0, xf]), torch.cat([meta0, metaf])
    drops = [None if a is None else torch.cat([a, b]) for a, b in zip(drops0, dropsf)]

    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    # the GPU's own B = 24 step
    out24 = run(model, x0, meta0, drops0, train=True)
    O.probe_loss(out24).backward()
    log24 = {t: v.detach().float().clone() for t, v in out24.items()}
    g24 = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    # the B = 256 step
    from linnaeus_amd import _lib as L
    persistent = lambda: L.lib().lnx_nt_kernel_launches(L.NT_KERNEL_V7) + L.lib().lnx_nt_kernel_launches(L.NT_KERNEL_V9)  # noqa: E731
    before = persistent()
    out = run(model, x, meta, drops, train=True)
    O.probe_loss({t: v[:B0] for t, v in out.items()}).backward()
    n_persistent = persistent() - before
    worst = worst24 = 0.0
    for t, _ in spec.heads:
        ref = oout[t].detach()
        got = out[t][:B0].detach().float().cpu()
        scale = max(1.0, ref.abs().max().item())
        err = (got - ref).abs().max().item() / scale
        err24 = (out[t][:B0].detach().float() - log24[t]).abs().max().item() / scale
        worst, worst24 = max(worst, err), max(worst24, err24)
        if dtype == "fp32":
            torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * scale, msg=t)
            assert (got.argmax(-1) == ref.argmax(-1)).all(), t
            assert err24 <= 1e-4, (t, err24)
        else:
            assert err <= 0.025, (t, err)
            assert err24 <= 0.012, (t, err24)  # two bf16 evaluations of the same samples on different tile grids / kernels
            srt = ref.sort(-1).values
            safe = (srt[:, -1] - srt[:, -2]) > 4 * err * scale
            assert (got.argmax(-1)[safe] == ref.argmax(-1)[safe]).all(), t
    glob, wk = _grad_errors(model, osd)
    e2 = r2 = 0.0
    for k, p_ in model.named_parameters():
        e2 += (p_.grad.double() - g24[k].double()).pow(2).sum().item()
        r2 += g24[k].double().pow(2).sum().item()
    glob24 = (e2 / r2) ** 0.5
    print(f"[sm B=256 rows 0-23 / {dtype}] logits vs oracle {worst:.5f}, vs the B=24 run {worst24:.5f} (of scale); gradient vs oracle {glob:.2e} "
          f"(worst {wk[0]} {wk[1]:.2e}), vs the B=24 run {glob24:.2e}; persistent NT launches in the step: {n_persistent}")
    assert glob <= (1e-3 if dtype == "fp32" else 5e-2), (glob, wk)
    assert glob24 <= (2e-4 if dtype == "fp32" else 3e-2), glob24
    if dtype == "bf16":
        # the forms the timed step gives to the persistent kernels: qkv (5), their data gradients (15), proj / fc2 with the fp32
        # residual (stage 3: 10), the GELU'-multiply data gradient (5 + 2)
        assert n_persistent >= 30, n_persistent


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config1_sm_b1_forward(dtype, golden_dir):
    """BASELINE config 1, literally: mFormerV1_sm forward, batch = 1, 3x224x224 + 5-wide metadata through build_model()
    (fixture-seeded weights, first sample of the sm fixture -> the reference's own logits are the expected values)."""
    spec, z, sd, x, meta, _ = load_case("sm", golden_dir)
    model = build("sm", spec, sd, dtype)
    with torch.no_grad():
        out = run(model, x[:1], meta[:1], None, train=False)
    for task, _ in spec.heads:
        ref = z["logits_" + task][:1]
        got = out[task].float().cpu().numpy()
        assert got.shape == ref.shape
        if dtype == "fp32":
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=5e-5, err_msg=task)
            assert (got.argmax(-1) == ref.argmax(-1)).all()
        else:
            assert np.abs(got - ref).max() <= 0.04 * max(1.0, np.abs(ref).max())


def test_config4_lg384_hierarchical_heads_taxonomy_loss():
    """BASELINE config 4 on one GPU: mFormerV1_lg @384, ConditionalClassifier heads on a 3-level taxonomy (SURVEY F3:
    effective shared Linear per task) and linnaeus_amd.loss.TaxonomyAwareLabelSmoothingCE per task (soft-label matrices,
    ignore_index 0, class weights), weighted sum of the per-task batch means -> backward.  Loss value, logits and
    every parameter gradient against the CPU oracle (O.forward + O.soft_label_ce)."""
    from linnaeus_amd.loss import TaxonomyAwareLabelSmoothingCE

    heads = (("taxa_L10", 48), ("taxa_L20", 12), ("taxa_L30", 4))
    spec = O.Spec(conv_dims=(192, 384, 768, 1536), rope_depths=(10, 2), rope_heads=(12, 24), meta=(("TEMPORAL", 2), ("SPATIAL", 3), ("ELEVATION", 10)),
                  heads=heads)
    B = 2
    sd = O.seeded_state_dict(O.param_shapes(spec), 4321)
    x, meta = O.seeded_inputs(spec, B, 384, 98)
    g = torch.Generator().manual_seed(5)
    soft, cw, tg, tw = {}, {}, {}, {"taxa_L10": 1.0, "taxa_L20": 0.5, "taxa_L30": 0.25}
    for t, c in heads:
        m = torch.rand(c, c, generator=g) * 0.1 + 0.9 * torch.eye(c)
        soft[t] = m / m.sum(1, keepdim=True)
        cw[t] = 0.5 + torch.rand(c, generator=g)
        tg[t] = torch.randint(0, c, (B,), generator=g)  # index 0 = null, ignored
    tg["taxa_L30"][0] = 0

    def total(out, dev):
        tot = 0.0
        for t, _ in heads:
            if dev == "cpu":
                per = O.soft_label_ce(out[t], tg[t], soft[t], cw[t], 0)
            else:
                crit = TaxonomyAwareLabelSmoothingCE(soft[t], weight=cw[t], apply_class_weights=True, ignore_index=0).cuda()
                per = crit(out[t], tg[t].cuda())
            tot = tot + tw[t] * per.mean()
        return tot

    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta)
    oloss = total(oout, "cpu")
    oloss.backward()
    tree = TinyTree({"taxa_L10": {i: i // 4 for i in range(48)}, "taxa_L20": {i: i // 3 for i in range(12)}}, [t for t, _ in heads], dict(heads))
    model = build_model(make_config(spec, 384, "ConditionalClassifier"), num_classes=dict(heads), taxonomy_tree=tree)
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    for dtype, ltol, ftol, gtol in (("fp32", 2e-4, 2e-4, 2e-3), ("bf16", 3e-2, 0.1, 0.1)):
        model.set_compute_dtype(dtype)
        model.train()
        model.zero_grad(set_to_none=True)
        out = model(x.cuda(), meta.cuda())
        for t, _ in heads:
            ref = oout[t].detach()
            err = (out[t].float().cpu() - ref).abs().max().item()
            assert err <= ftol * max(1.0, ref.abs().max().item()), (dtype, t, err)
        loss = total(out, "cuda")
        assert abs(loss.item() - oloss.item()) <= ltol * max(1.0, abs(oloss.item())), (dtype, loss.item(), oloss.item())
        loss.backward()
        glob, wk = _grad_errors(model, osd)
        print(f"[lg@384 hierarchical+taxonomic/{dtype}] loss {loss.item():.5f} vs {oloss.item():.5f}; global relative gradient error {glob:.2e}")
        assert glob <= gtol, (dtype, glob, wk)


# ----------------------------------------------------------------------------------------------------
# Round 5: configs 4 and 5 at the dispatch their bench lines time -- batch invariance, as
# test_sm_b256_batch_invariance does for config 2.  Reference math of the kernels this reaches: rope_2d_mhsa.py:422-505,
# blocks/mlp.py:46-66, blocks/convnext.py:73-87 (every sample independent: mFormerV1.py:407-541 has no batch statistics).
# ----------------------------------------------------------------------------------------------------
def _nt_counts():
    from linnaeus_amd import _lib as L

    return {k: L.lib().lnx_nt_kernel_launches(getattr(L, "NT_KERNEL_" + k)) for k in ("V1", "V2", "V4", "V7", "V9", "FP8", "MX8")}


def _rel(ga, gb):
    e2 = sum((ga[k].double() - gb[k].double()).pow(2).sum().item() for k in gb)
    r2 = sum(gb[k].double().pow(2).sum().item() for k in gb)
    return (e2 / r2) ** 0.5


def _batch_invariance_case(model, osd, oout, x0, meta0, drops0, xf, metaf, dropsf, heads, loss_of, dtype, ftol, gtol, ftol_self, gtol_self, tag):
    """Rows 0..B0-1 of the big batch are the oracle-checked small-batch inputs (same DropPath multipliers), the rest seeded filler.
    (a) logits of those rows against the oracle and against the GPU's own small-batch run; (b) with the loss restricted to those rows
    every parameter gradient against the oracle's and the small-batch run's.  Returns the NT dispatcher's launch counts of the big step."""
    B0 = x0.shape[0]
    model.set_compute_dtype(dtype)
    model.zero_grad(set_to_none=True)
    out_s = run(model, x0, meta0, drops0, train=True)
    loss_of(out_s, "cuda").backward()
    log_s = {t: v.detach().float().clone() for t, v in out_s.items()}
    g_s = _grads(model)
    model.zero_grad(set_to_none=True)
    x, meta = torch.cat([x0, xf]), torch.cat([meta0, metaf])
    drops = [None if a is None else torch.cat([a, b]) for a, b in zip(drops0, dropsf)]
    before = _nt_counts()
    out = run(model, x, meta, drops, train=True)
    loss_of({t: v[:B0] for t, v in out.items()}, "cuda").backward()
    torch.cuda.synchronize()
    after = _nt_counts()
    worst = worst_s = 0.0
    for t, _ in heads:
        ref = oout[t].detach()
        got = out[t][:B0].detach().float().cpu()
        scale = max(1.0, ref.abs().max().item())
        worst = max(worst, (got - ref).abs().max().item() / scale)
        worst_s = max(worst_s, (out[t][:B0].detach().float() - log_s[t]).abs().max().item() / scale)
        assert torch.isfinite(out[t]).all(), t
    glob, wk = _grad_errors(model, osd)
    glob_s = _rel(_grads(model), g_s)
    print(f"[{tag} rows 0-{B0 - 1} of {x.shape[0]} / {dtype}] logits vs oracle {worst:.5f}, vs the small-batch run {worst_s:.5f} (of scale); gradient vs oracle "
          f"{glob:.2e} (worst {wk[0]} {wk[1]:.2e}), vs the small-batch run {glob_s:.2e}")
    assert worst <= ftol, (tag, dtype, worst)
    assert worst_s <= ftol_self, (tag, dtype, worst_s)
    assert glob <= gtol, (tag, dtype, glob, wk)
    assert glob_s <= gtol_self, (tag, dtype, glob_s)
    model.zero_grad(set_to_none=True)
    return {k: after[k] - before[k] for k in after}


def test_xl_b128_batch_invariance():
    """BASELINE config 5 at the batch its bench lines time (`bench.py --arch xl --batch 128`, bf16 and --dtype fp8): mFormerV1_xl @224,
    B = 128, DropPath 0.2.  M = 128 x 199 = 25 472 rows in stage 3: the persistent 256x256 kernel gemm_nt_v9 on 400-tile grids at
    (N, K) in {(1024,1024), (3072,1024), (4096,1024), (1024,4096), (1024,3072)}, gemm_tn_v2 at those widths, the unfused C = 256 / 512 conv
    path with the GELU' / multiply epilogues, and in fp8 mode gemm_nt_mx8_kernel inside the plan (its counter proves it).  Rows 0-1 are the
    inputs of an oracle run.  Stated tolerances (fractions of the logit scale / of the global gradient norm): bf16 0.04 / 0.05 against the oracle as in
    test_xlarge_224_matches_oracle_and_autobatch (measured here 0.008 / 0.012) and 0.02 / 0.03 against the GPU's own B = 2 run (measured 0.004 / 0.009);
    fp8 (MXFP8 forward products AND data gradients, the mode's default since round 5) 0.16 / 0.14 against the oracle (measured 0.105 / 0.109 with
    these DropPath draws: 22 MXFP8 blocks deep on random-init weights; 0.086 with bf16 data gradients) and 0.07 / 0.11 against the B = 2 fp8 run
    (measured 0.048 / 0.079: the two batches take different bf16 attention / GEMM tilings, and a bf16 rounding flip in a LayerNorm output or a dY
    moves an e4m3 quantisation step of 6 % -- the mode amplifies the bf16 mode's 0.004 / 0.009 about tenfold, as it does the error itself)."""
    spec = O.Spec(conv_dims=(256, 512, 1024, 2048), rope_depths=(22, 2), rope_heads=(16, 32), heads=(("taxa_L10", 40), ("taxa_L20", 9)), drop_path_rate=0.2)
    B0, B = 2, 128
    sd = O.seeded_state_dict(O.param_shapes(spec), 4321)
    x0, meta0 = O.seeded_inputs(spec, B0, 224, 77)
    drops0 = _drop_scales(spec, B0, 78)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x0, meta0, drops0)
    O.probe_loss(oout).backward()
    xf, metaf = O.seeded_inputs(spec, B - B0, 224, 79)
    dropsf = _drop_scales(spec, B - B0, 80)
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    loss_of = lambda out, dev: O.probe_loss(out)  # noqa: E731
    n_rope3 = 22
    for dtype, ftol, gtol, ftol_self, gtol_self in (("bf16", 0.04, 0.05, 0.02, 0.03), ("fp8", 0.16, 0.14, 0.07, 0.11)):
        d = _batch_invariance_case(model, osd, oout, x0, meta0, drops0, xf, metaf, dropsf, spec.heads, loss_of, dtype, ftol, gtol, ftol_self, gtol_self, "xl@224")
        print(f"[xl@224 B=128 / {dtype}] NT launches in the step: {d}")
        if dtype == "bf16":
            # stage 3, per block: qkv, proj, fc1, fc2 and the proj / fc1 / qkv data gradients on the persistent 256x256 kernel (N % 256 == 0, 400+
            # tiles); the fc2 data gradient (multiply-by-GELU' epilogue) on the persistent 256x128 kernel (gemm2.hip: nt_v7_preferred)
            assert d["V9"] >= 7 * n_rope3 and d["V7"] + d["V9"] >= 8 * n_rope3, d
            assert d["MX8"] == 0 and d["FP8"] == 0, d
        else:
            # the MXFP8 forward products (qkv, fc1, fc2) and data gradients (proj, fc1, fc2) of the 22 stage-3 blocks on the 256x256 MX kernel; the
            # proj forward and the qkv data gradient stay bf16 (gemm_nt_v9)
            assert d["MX8"] >= 6 * n_rope3, d
            assert d["V9"] >= 2 * n_rope3, d


def _config4_case(B):
    """BASELINE config 4's pieces: mFormerV1_lg @384 with three metadata components (N = 24 x 24 + 4 = 580 tokens), ConditionalClassifier heads
    on a 3-level taxonomy and TaxonomyAwareLabelSmoothingCE per task (soft-label matrices, ignore_index 0, class weights)."""
    from linnaeus_amd.loss import TaxonomyAwareLabelSmoothingCE

    heads = (("taxa_L10", 48), ("taxa_L20", 12), ("taxa_L30", 4))
    g = torch.Generator().manual_seed(5)
    soft, cw, tg, tw = {}, {}, {}, {"taxa_L10": 1.0, "taxa_L20": 0.5, "taxa_L30": 0.25}
    for t, c in heads:
        m = torch.rand(c, c, generator=g) * 0.1 + 0.9 * torch.eye(c)
        soft[t] = m / m.sum(1, keepdim=True)
        cw[t] = 0.5 + torch.rand(c, generator=g)
        tg[t] = torch.randint(0, c, (B,), generator=g)  # index 0 = null, ignored
    tg["taxa_L30"][0] = 0
    crits = {}

    def total(out, dev):
        tot = 0.0
        for t, _ in heads:
            if dev == "cpu":
                per = O.soft_label_ce(out[t], tg[t], soft[t], cw[t], 0)
            else:
                if t not in crits:
                    crits[t] = TaxonomyAwareLabelSmoothingCE(soft[t], weight=cw[t], apply_class_weights=True, ignore_index=0).cuda()
                per = crits[t](out[t], tg[t].cuda())
            tot = tot + tw[t] * per.mean()
        return tot

    tree = TinyTree({"taxa_L10": {i: i // 4 for i in range(48)}, "taxa_L20": {i: i // 3 for i in range(12)}}, [t for t, _ in heads], dict(heads))
    return heads, total, tree


def test_lg384_b64_batch_invariance():
    """BASELINE config 4 at the batch its bench line times (`bench.py --arch lg --img 384 --batch 64`): mFormerV1_lg @384, B = 64, DropPath 0.2,
    ConditionalClassifier heads + taxonomy-aware label-smoothing loss.  M = 64 x 580 = 37 120 rows in stage 3: gemm_nt_v9 at (N, K) in
    {(768,768), (2304,768), (3072,768), (768,3072), (768,2304)}, the multi-tile attention path at N = 580, the C = 192 fused and C = 384 unfused
    conv blocks at 96 x 96 / 48 x 48.  Rows 0-1 are the inputs of an oracle run (loss value included); bounds as test_config4_lg384_... / test_large_384_..."""
    B0, B = 2, 64
    heads, total, tree = _config4_case(B0)
    spec = O.Spec(conv_dims=(192, 384, 768, 1536), rope_depths=(10, 2), rope_heads=(12, 24), meta=(("TEMPORAL", 2), ("SPATIAL", 3), ("ELEVATION", 10)),
                  heads=heads, drop_path_rate=0.2)
    sd = O.seeded_state_dict(O.param_shapes(spec), 4321)
    x0, meta0 = O.seeded_inputs(spec, B0, 384, 98)
    drops0 = _drop_scales(spec, B0, 97)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x0, meta0, drops0)
    oloss = total(oout, "cpu")
    oloss.backward()
    xf, metaf = O.seeded_inputs(spec, B - B0, 384, 96)
    dropsf = _drop_scales(spec, B - B0, 95)
    model = build_model(make_config(spec, 384, "ConditionalClassifier"), num_classes=dict(heads), taxonomy_tree=tree)
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    seen = {}

    def loss_of(out, dev):
        l_ = total(out, dev)
        seen["loss"] = float(l_.detach())
        return l_

    d = _batch_invariance_case(model, osd, oout, x0, meta0, drops0, xf, metaf, dropsf, heads, loss_of, "bf16", 0.03, 0.05, 0.02, 0.03, "lg@384")
    print(f"[lg@384 B=64 / bf16] loss on rows 0-1 {seen['loss']:.5f} vs oracle {oloss.item():.5f}; NT launches in the step: {d}")
    assert abs(seen["loss"] - oloss.item()) <= 3e-2 * max(1.0, abs(oloss.item())), (seen["loss"], oloss.item())
    # stage 3 (10 blocks): every product has N % 256 == 0 and 145 x 3 = 435 tiles or more; the fc2 data gradient takes gemm_nt_v7
    assert d["V9"] >= 7 * 10 and d["V7"] + d["V9"] >= 8 * 10, d


def test_plan_cache_is_bounded_and_released():
    """every distinct batch size used to pin another workspace forever.  The plan cache is an LRU of
    `max_cached_plans`; evicted plans are destroyed and their workspaces freed; no_grad forwards use an inference plan
    without backward scratch."""
    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 3)
    model = build("tiny_a", spec, sd, "bf16")
    model.max_cached_plans = 2
    model.eval()
    torch.cuda.synchronize()
    base = None
    peak = 0
    with torch.no_grad():
        for B in (1, 2, 3, 4, 5, 6, 7, 8, 3, 9, 2):
            x, meta = O.seeded_inputs(spec, B, 64, 50 + B)
            model(x.cuda(), meta.cuda())
            torch.cuda.synchronize()
            assert len(model._plans) <= 2
            if base is None:
                base = torch.cuda.memory_allocated()
            peak = max(peak, torch.cuda.memory_allocated())
    # workspace grows ~linearly in B: bounded by (two largest plans) rather than the sum over all 9 sizes
    assert peak - base < 20 * base + (64 << 20)
    model.train()
    x, meta = O.seeded_inputs(spec, 2, 64, 1)
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()  # a training plan after inference plans of the same shape
    assert all(p_.grad is not None for p_ in model.parameters())
    model.release_plans()
    assert len(model._plans) == 0


def test_switch_autograd_to_direct_grad_mode():
    """after an autograd-mode step + zero_grad(set_to_none=False), direct mode must not accumulate onto a
    stale arena."""
    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 1)
    model = build("tiny_a", spec, sd, "fp32")
    x, meta = O.seeded_inputs(spec, 2, 64, 5)
    model.train()
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()  # autograd mode: .grad are clones
    want = {k: p_.grad.clone() for k, p_ in model.named_parameters()}
    model.zero_grad(set_to_none=False)
    model.grad_mode = "direct"
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()
    for k, p_ in model.named_parameters():
        torch.testing.assert_close(p_.grad, want[k], rtol=1e-4, atol=1e-6, msg=k)
    O.probe_loss(model(x.cuda(), meta.cuda())).backward()  # now accumulates (grads alias the arena)
    for k, p_ in model.named_parameters():
        torch.testing.assert_close(p_.grad, 2 * want[k], rtol=1e-4, atol=1e-6, msg=k)


def test_autobatch_analytic_matches_measured():
    """AutoBatch (utils/autobatch.py:111-265): the planner's analytic footprint predicts the measured peak of a real
    training step within a few percent, the search result is the largest batch under the budget, and a tight budget
    yields a smaller batch than a loose one."""
    import gc

    from linnaeus_amd.autobatch import _trial, auto_find_batch_size, foreign_bytes, predicted_bytes

    gc.collect()
    torch.cuda.empty_cache()

    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 9)
    model = build("tiny_a", spec, sd, "bf16")
    cfg = make_config(spec, 64)
    for B in (8, 64):
        pred = predicted_bytes(model, B, 64, mode="train", optimizer_state_per_param=0)
        base = torch.cuda.memory_allocated()
        peak = _trial(model, B, 64, "train", 2, None) - base + 4 * sum(p_.numel() for p_ in model.parameters())
        assert 0.7 * pred <= peak <= 1.3 * pred + (8 << 20), (B, pred, peak)
    total = torch.cuda.get_device_properties(0).total_memory
    per_img = (predicted_bytes(model, 512, 64) - predicted_bytes(model, 256, 64)) / 256
    other = foreign_bytes(model)  # tensors other tests of this process still hold count against the budget
    frac = (predicted_bytes(model, 300, 64) + 0.5 * per_img + other) / total
    bs = auto_find_batch_size(model, cfg, "train", target_memory_fraction=frac, max_batch_size=4096, min_batch_size=1, steps_per_trial=1)
    assert 280 <= bs <= 300, bs
    assert predicted_bytes(model, bs, 64) + other <= frac * total < predicted_bytes(model, bs + 2, 64) + other + per_img
    half = (predicted_bytes(model, 150, 64) + other) / total
    assert auto_find_batch_size(model, cfg, "train", target_memory_fraction=half, max_batch_size=4096, steps_per_trial=1) < bs


def test_opt_in_hierarchical_refinement(golden_dir):
    """MODEL.CLASSIFICATION.HIERARCHICAL_REFINEMENT (opt-in, off by default): refined[child] = base[child] +
    log(softmax(refined[parent]) . M + 1e-10), coarsest rank first.  The reference never executes this (F3/F4), so the
    expected values are the formula itself in fp64 on the base logits -- parity for this flag is unpinned by design."""
    spec, z, sd, x, meta, drops = load_case("tiny_c", golden_dir)
    model = build("tiny_c", spec, sd, "fp32")
    model.eval()
    with torch.no_grad():
        base = {k: v.clone() for k, v in model(x.cuda(), None).items()}
        model.hierarchical_refinement = True
        got = model(x.cuda(), None)
    keys = [t for t, _ in spec.heads]
    ref = {k: v.double() for k, v in base.items()}
    for i in range(len(keys) - 2, -1, -1):
        child, parent = keys[i], keys[i + 1]
        M = getattr(model.head[child], f"hmatrix_{parent}_{child}").double()
        ref[child] = base[child].double() + torch.log(torch.softmax(ref[parent], 1) @ M + 1e-10)
    for k in keys:
        torch.testing.assert_close(got[k].double(), ref[k], rtol=1e-5, atol=1e-5)
    assert not torch.allclose(got[keys[0]], base[keys[0]])  # the flag does change the fine ranks
    torch.testing.assert_close(got[keys[-1]], base[keys[-1]])  # the coarsest rank has no parent
    model.train()
    out = model(x.cuda(), None)
    O.probe_loss(out).backward()  # differentiable through the HIP GEMM of the prior
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in model.parameters())


# ----------------------------------------------------------------------------------------------------
# A16: activation recompute (gradient checkpointing) -- convnext.py:89-100, rope_2d_mhsa.py:617-641
# ----------------------------------------------------------------------------------------------------
def _grads(model):
    return {k: p_.grad.detach().clone() for k, p_ in model.named_parameters()}


@pytest.mark.parametrize("name,dtype", [("tiny_dp", "fp32"), ("tiny_b", "bf16"), ("tiny_c", "fp32")])
def test_recompute_plan_equals_kept_activations(name, dtype, golden_dir):
    """forward(force_checkpointing=True) in training mode runs a recompute plan: same logits bit for bit, same gradients
    (up to the summation order of the atomically accumulated ones), smaller workspace; it is ignored in eval mode, as in
    the reference's blocks (`use_checkpoint and self.training`)."""
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    model = build(name, spec, sd, dtype)
    model.train(True)
    model._inject_drop = drops
    xs, ms = x.cuda(), meta.cuda() if meta is not None else None
    out_a = model(xs, ms)
    O.probe_loss(out_a).backward()
    ga = _grads(model)
    model.zero_grad(set_to_none=True)
    out_b = model(xs, ms, force_checkpointing=True)
    assert model._active["handle"] is not None and any(k[-1] for k in model._plans), "no recompute plan was created"
    for t in out_a:
        assert torch.equal(out_a[t], out_b[t]), t
    O.probe_loss(out_b).backward()
    gb = _grads(model)
    for k in ga:
        err = (ga[k] - gb[k]).norm().item()
        assert err <= 1e-5 * ga[k].norm().item() + 1e-7, (k, err, ga[k].norm().item())
    B = x.shape[0]
    deep = max(spec.conv_depths + spec.rope_depths) > 1   # one block per stage: nothing to share
    w_ck, w_keep = model.workspace_bytes(B, IMG[name], recompute=True), model.workspace_bytes(B, IMG[name], recompute=False)
    assert w_ck < w_keep if deep else w_ck == w_keep, (w_ck, w_keep)
    # oracle check of the recompute path itself (fp32 cases)
    if dtype == "fp32":
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        O.probe_loss(O.forward(osd, spec, x, meta, drops)).backward()
        glob, wk = _grad_errors(model, osd)
        assert glob <= 1e-3, (glob, wk)
    # eval mode: the flag is ignored (no recompute plan for an eval forward under grad)
    n_plans = len(model._plans)
    model.train(False)
    model(xs, ms, force_checkpointing=True)
    assert not any(k[-1] for k in list(model._plans)[n_plans:])


@pytest.mark.parametrize("dtype,B", [("bf16", 64), ("fp32", 24)])
def test_weight_gradient_stream_changes_no_gradient(dtype, B):
    """The backward's weight-gradient stream (include/lnx.h lnx_plan_set_wgrad_stream; on by default): sm @224 at the production dispatch,
    one training step with the stream on and one with everything on the launch stream, same forward.  The products, their split-K
    partial sums and the order they are added in are the same either way, so every gradient whose reduction order is fixed must be
    bit-equal (asserted for the RoPE blocks' Linear weights, counted for the rest) -- a missing join (a weight-gradient product
    reading a dY buffer its next writer already reached) or a block's gradients handed on before its stream finished shows up as a
    large difference, and twice over two backwards of one forward.  Gradients that accumulate by atomics (sparsely filled
    weight-gradient tiles, LayerNorm partial sums, the metadata-head chains on their side stream) compare to 1e-5."""
    from linnaeus_amd import _lib as L

    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300)), drop_path_rate=0.2)
    sd = O.seeded_state_dict(O.param_shapes(spec), 41)
    x, meta = O.seeded_inputs(spec, B, 224, 42)
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    model.train(True)
    model._inject_drop = _drop_scales(spec, B, 43)
    xs, ms = x.cuda(), meta.cuda()
    out = model(xs, ms)
    loss = O.probe_loss(out)
    handle = model._active["handle"]
    was = L.lib().lnx_plan_set_wgrad_stream(handle, 1)
    assert was == 1, "the weight-gradient stream is on by default"
    loss.backward(retain_graph=True)
    g_on = _grads(model)
    model.zero_grad(set_to_none=True)
    assert L.lib().lnx_plan_set_wgrad_stream(handle, 0) == 1
    try:
        loss.backward(retain_graph=True)
        g_off = _grads(model)
        model.zero_grad(set_to_none=True)
    finally:
        assert L.lib().lnx_plan_set_wgrad_stream(handle, 1) == 0
    loss.backward()
    g_on2 = _grads(model)
    exact = 0
    for k in g_off:
        for g in (g_on[k], g_on2[k]):
            assert (g - g_off[k]).norm().item() <= 1e-5 * g_off[k].norm().item() + 1e-7, (k, (g - g_off[k]).abs().max().item())
        same = torch.equal(g_on[k], g_off[k]) and torch.equal(g_on2[k], g_off[k])
        exact += same
        # the RoPE blocks' Linear weights in bf16: well-filled split-K tiles through the workspace, summed in split order by the batched
        # reduce (the fp32 plans' weight-gradient kernel adds its splits by atomics: order not fixed, 5e-10 apart)
        if dtype == "bf16" and k.startswith("stages.2.") and k.endswith((".qkv.weight", ".proj.weight", ".fc1.weight", ".fc2.weight")):
            assert same, (k, (g_on[k] - g_off[k]).abs().max().item())
    assert dtype != "bf16" or exact >= len(g_off) // 2, (exact, len(g_off))
    assert L.lib().lnx_plan_set_wgrad_stream(handle, 2) < 0  # rejected, setting unchanged
    assert L.lib().lnx_plan_set_wgrad_stream(handle, 1) == 1
    # the module-level switch reaches the plans that exist and the ones created later
    model.set_wgrad_stream(False)
    assert L.lib().lnx_plan_set_wgrad_stream(handle, 0) == 0
    model._inject_drop = _drop_scales(spec, 8, 44)
    model(xs[:8], ms[:8])  # another batch size: a new plan
    h2 = model._active["handle"]
    assert h2 is not handle and L.lib().lnx_plan_set_wgrad_stream(h2, 0) == 0
    model.set_wgrad_stream(True)
    assert L.lib().lnx_plan_set_wgrad_stream(h2, 1) == 1 and L.lib().lnx_plan_set_wgrad_stream(handle, 1) == 1


def test_recompute_sm_b24_production_dispatch_and_config_flag():
    """The recompute plan at the production dispatch (sm@224, B = 24, bf16, DropPath on), selected the way the reference's
    train loop selects it (TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS / model.use_checkpoint, train.py:93-110),
    run segment by segment as DataParallel runs it, and backward twice over one forward (the second pass has to restore
    the last block of each stage as well)."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300)), drop_path_rate=0.2)
    B = 24
    sd = O.seeded_state_dict(O.param_shapes(spec), 31)
    x, meta = O.seeded_inputs(spec, B, 224, 32)
    drops = _drop_scales(spec, B, 33)
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype("bf16")
    model.train(True)
    model._inject_drop = drops
    xs, ms = x.cuda(), meta.cuda()
    out_a = model(xs, ms)
    O.probe_loss(out_a).backward()
    ga = _grads(model)
    kept_ws = model._active["ws"].numel()
    model.zero_grad(set_to_none=True)

    model.use_checkpoint = True           # what train.py sets from the config flag
    seen = []
    model._segment_hook = seen.append     # segment-wise backward, as under DataParallel
    out_b = model(xs, ms)
    assert model._active["ws"].numel() < 0.75 * kept_ws, (model._active["ws"].numel(), kept_ws)
    for t in out_a:
        assert torch.equal(out_a[t], out_b[t]), t
    loss = O.probe_loss(out_b)
    loss.backward(retain_graph=True)
    assert seen == [0, 1, 2, 3]
    gb = _grads(model)
    for k in ga:
        err = (ga[k] - gb[k]).norm().item()
        assert err <= 1e-5 * ga[k].norm().item() + 1e-7, (k, err, ga[k].norm().item())
    # a second backward over the same forward accumulates the same gradient again
    model._segment_hook = None
    loss.backward()
    for k, p_ in model.named_parameters():
        err = (p_.grad - 2 * ga[k]).norm().item()
        assert err <= 2e-5 * ga[k].norm().item() + 2e-7, (k, err)


def test_fused_block_layernorm_matches_separate_passes(monkeypatch):
    """sm@224, B = 8, bf16, DropPath on: the plan with the conv-block LayerNorm inside the fused conv-MLP kernels (default) against the
    plan that runs it as its own forward / backward passes (LNX_NO_FUSED_LN, read when a plan is created).  Same arithmetic up to the
    summation order of the row statistics: logits agree to bf16 rounding noise, gradients to 1 % globally -- and both paths really are
    different plans (the LayerNorm gradients are not bit-equal)."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300)), drop_path_rate=0.2)
    B = 8
    sd = O.seeded_state_dict(O.param_shapes(spec), 41)
    x, meta = O.seeded_inputs(spec, B, 224, 42)
    drops = _drop_scales(spec, B, 43)

    def run():
        model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
        model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
        model = model.cuda()
        model.set_compute_dtype("bf16")
        model.train(True)
        model._inject_drop = drops
        out = model(x.cuda(), meta.cuda())
        O.probe_loss(out).backward()
        return {t: v.detach().float() for t, v in out.items()}, _grads(model)

    out_f, g_f = run()
    monkeypatch.setenv("LNX_NO_FUSED_LN", "1")
    out_s, g_s = run()
    for t in out_f:
        scale = out_s[t].abs().max().item()
        assert (out_f[t] - out_s[t]).abs().max().item() <= 0.02 * scale, t
    num = sum((g_f[k].double() - g_s[k].double()).pow(2).sum().item() for k in g_f) ** 0.5
    den = sum(g_s[k].double().pow(2).sum().item() for k in g_s) ** 0.5
    assert num <= 0.01 * den, (num, den)
    ln_keys = [k for k in g_f if k.startswith("stages.0.") and ".norm." in k]
    assert ln_keys and any(not torch.equal(g_f[k], g_s[k]) for k in ln_keys), "both runs took the same path"
    for k in ln_keys:  # the gradients the fused backward produces itself
        err = (g_f[k] - g_s[k]).norm().item()
        assert err <= 0.02 * g_s[k].norm().item() + 1e-6, (k, err, g_s[k].norm().item())


# ----------------------------------------------------------------------------------------------------
# (f-4) / config 5: the fp8 MFMA path at the model level (MXFP8 forward products in the RoPE blocks)
# ----------------------------------------------------------------------------------------------------
def test_fp8_mode_sm_b24_against_oracle_and_bf16():
    """compute dtype 'fp8' = bf16 plan + MXFP8 (block-scaled e4m3) qkv / fc1 / fc2 forward products.  Stated tolerance against
    the fp32 oracle (the reference has no fp8 code): logits within 0.12 x their scale (bf16 mode: 0.04), arg-max equal
    wherever the oracle's margin exceeds 4x the error, global relative gradient error <= 12 % (bf16 mode: 5 %) -- the
    gradients are bf16 products evaluated at the fp8 forward's activations.  Also: the fp8 plan really runs the fp8
    kernels (its logits differ from the bf16 plan's), and eval / recompute plans work in this mode."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20)), drop_path_rate=0.2)
    B = 24
    sd = O.seeded_state_dict(O.param_shapes(spec), 777)
    x, meta = O.seeded_inputs(spec, B, 224, 778)
    drops = _drop_scales(spec, B, 779)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta, drops)
    O.probe_loss(oout).backward()
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype("bf16")
    out_bf = {t: v.detach().clone() for t, v in run(model, x, meta, drops, train=True).items()}
    model.set_compute_dtype("fp8")
    assert model.compute_dtype == "fp8"
    out = run(model, x, meta, drops, train=True)
    worst = 0.0
    for t, _ in spec.heads:
        ref = oout[t].detach()
        got = out[t].float().cpu()
        err = (got - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item())
        worst = max(worst, err / scale)
        assert err <= 0.12 * scale, (t, err, scale)
        srt = ref.sort(-1).values
        safe = (srt[:, -1] - srt[:, -2]) > 4 * err
        assert (got.argmax(-1)[safe] == ref.argmax(-1)[safe]).all(), t
        assert not torch.equal(out[t], out_bf[t]), "the fp8 plan produced the bf16 plan's logits: fp8 kernels not in use"
    O.probe_loss(out).backward()
    glob, wk = _grad_errors(model, osd)
    print(f"[sm B=24/fp8] max logit error / scale {worst:.4f}; global relative gradient error {glob:.3e}; worst tensor {wk[0]} {wk[1]:.2e}")
    assert glob <= 0.12, (glob, wk)
    # eval (inference plan) and recompute plan in fp8 mode
    model.zero_grad(set_to_none=True)
    ev = run(model, x, meta, None, train=False)
    model.train(True)
    model._inject_drop = [None] * len(drops)
    tr = model(x.cuda(), meta.cuda(), force_checkpointing=True)
    for t in ev:
        assert torch.equal(ev[t], tr[t]), t
    O.probe_loss(tr).backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in model.parameters())


def test_fp8_dgrad_opt_in(monkeypatch):
    """fp8 plans run the proj / fc2 / fc1 data-gradient products in MXFP8 too (transposed MXFP8 weight copies, dY quantised per branch, dH
    handed on by the GELU' epilogue) -- the default since round 5, LNX_FP8_DGRAD=0 keeps them bf16.  Same forward either way; gradients
    within the fp8 mode's stated 12 % of the oracle in both."""
    spec = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300)), drop_path_rate=0.2)
    B = 24
    sd = O.seeded_state_dict(O.param_shapes(spec), 777)
    x, meta = O.seeded_inputs(spec, B, 224, 778)
    drops = _drop_scales(spec, B, 779)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.probe_loss(O.forward(osd, spec, x, meta, drops)).backward()
    model = build_model(make_config(spec, 224), num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype("fp8")
    monkeypatch.setenv("LNX_FP8_DGRAD", "0")
    out_a = run(model, x, meta, drops, train=True)
    O.probe_loss(out_a).backward()
    glob_bf, _ = _grad_errors(model, osd)
    g_bf = _grads(model)
    out_a = {t: v.detach().clone() for t, v in out_a.items()}
    model.zero_grad(set_to_none=True)
    monkeypatch.delenv("LNX_FP8_DGRAD")
    model.release_plans()
    out = run(model, x, meta, drops, train=True)
    for t in out:
        assert torch.equal(out[t], out_a[t]), t
    O.probe_loss(out).backward()
    glob, wk = _grad_errors(model, osd)
    print(f"[sm B=24/fp8] global relative gradient error: MXFP8 data gradients (default) {glob:.3e}, bf16 data gradients {glob_bf:.3e}; worst tensor {wk[0]} {wk[1]:.2e}")
    assert glob <= 0.12 and glob_bf <= 0.12, (glob, glob_bf, wk)
    assert _rel(_grads(model), g_bf) > 1e-4, "both settings gave the same gradients: the MXFP8 data-gradient products did not run"
    # round 4: dY reaches the fc2 / proj data-gradient products as the MXFP8 copy the LayerNorm backward wrote beside its bf16 output; with
    # LNX_FP8_DGRAD_QPASS it is quantised by a separate pass over that bf16 tensor instead -- the same bytes
    # (test_layernorm_bwd_second_output_and_its_mxfp8_copy), hence the same gradients up to what two backward passes differ by anyway
    # (the atomics of DESIGN 8b': 1e-9-level differences in the stream gradient reach every tensor)
    g_fused = _grads(model)
    model.zero_grad(set_to_none=True)
    monkeypatch.setenv("LNX_FP8_DGRAD_QPASS", "1")
    out_q = run(model, x, meta, drops, train=True)
    O.probe_loss(out_q).backward()
    g_pass = _grads(model)
    for k_ in g_pass:
        err = (g_fused[k_] - g_pass[k_]).norm().item()
        assert err <= 1e-5 * g_pass[k_].norm().item() + 1e-7, (k_, err)


def test_fp8_mode_rejects_unsupported_widths(golden_dir):
    """MXFP8 blocks are 32 wide and a K slice of the kernel 128: the planner refuses other widths loudly (no silent bf16)."""
    import ctypes as C
    from linnaeus_amd import _lib as L

    spec, z, sd, x, meta, drops = load_case("tiny_a", golden_dir)
    model = build("tiny_a", spec, sd, "bf16")
    model.set_compute_dtype("fp8")
    cfg = model._make_cfg(2, 64, 64, True)
    assert cfg.fp8 == 1
    cfg.mlp_hidden[0] = 192
    handle = C.c_void_p()
    with pytest.raises(L.LnxError, match="multiples of 128"):
        L.check(L.lib().lnx_plan_create(C.byref(cfg), C.byref(handle)), "lnx_plan_create")


def test_fp8_plan_with_a_narrow_rope_stage_falls_back_to_bf16():
    """RoPE dims that are multiples of 128 but below the MXFP8 kernel's K >= 256 floor (tiny: C = 128 with
    16 x 19 = 304 >= 256 rows) used to create a plan that then failed inside lnx_plan_forward; such a stage now runs its bf16
    products, like a stage with fewer than 256 rows (C = 256 with 112 rows here): fp8 mode == bf16 mode, bit for bit."""
    spec = CASES["tiny_a"]
    sd = O.seeded_state_dict(O.param_shapes(spec), 5)
    x, meta = O.seeded_inputs(spec, 16, 64, 6)
    outs, grads = {}, {}
    for dtype in ("bf16", "fp8"):
        model = build("tiny_a", spec, sd, dtype)
        model.train()
        out = model(x.cuda(), meta.cuda())
        O.probe_loss(out).backward()
        outs[dtype] = {t: v.detach().clone() for t, v in out.items()}
        grads[dtype] = {k: p_.grad.clone() for k, p_ in model.named_parameters()}
    for t in outs["bf16"]:
        assert torch.equal(outs["bf16"][t], outs["fp8"][t]), t
    rel = sum((grads["fp8"][k] - grads["bf16"][k]).double().pow(2).sum().item() for k in grads["bf16"]) ** 0.5
    rel /= sum(grads["bf16"][k].double().pow(2).sum().item() for k in grads["bf16"]) ** 0.5
    assert rel < 1e-5, rel  # same kernels; the float atomics of the weight-gradient sums are the only run-to-run difference


def test_backward_through_features_only_and_a_single_task(golden_dir):
    """autograd may reach the model with gradients for only some of its outputs: forward_features() alone (no logits
    gradient at all) and a loss on one task (the other heads then get exactly zero gradient, the trunk only that task's)."""
    spec, z, sd, x, meta, drops = load_case("tiny_b", golden_dir)
    model = build("tiny_b", spec, sd, "fp32")
    model.train(True)
    model._inject_drop = drops
    xs, ms = x.cuda(), meta.cuda() if meta is not None else None
    feats = model.forward_features(xs, ms)
    feats.square().sum().backward()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.forward_features(osd, spec, x, meta, drops).square().sum().backward()
    for k, p_ in model.named_parameters():
        if k.startswith("head."):
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, k
        else:
            ref = osd[k].grad
            assert (p_.grad.cpu() - ref).norm() <= 2e-3 * max(ref.norm().item(), 1e-3), k
    model.zero_grad(set_to_none=True)
    out = model(xs, ms)
    t0, t1 = list(out)[:2]
    out[t0].float().square().sum().backward()
    for v in osd.values():
        v.grad = None
    O.forward(osd, spec, x, meta, drops)[t0].square().sum().backward()
    for k, p_ in model.named_parameters():
        ref = osd[k].grad if k in osd else None
        if k.startswith("head.") and t1 in k:
            assert float(p_.grad.abs().max()) == 0.0, k
        elif ref is not None:
            assert (p_.grad.cpu() - ref).norm() <= 2e-3 * max(ref.norm().item(), 1e-3), k


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_grad_scaler_flow_of_the_reference_train_loop(dtype, golden_dir):
    """train.py:176,280,311-312: scaler.scale(loss).backward(); scaler.unscale_(optimizer); clip_grad_norm_; scaler.step();
    scaler.update() under torch.autocast(float16) around the drop-in model (A18).  The loss scale is a power of two, so the
    step must equal the unscaled flow's (bf16 operands share fp32's exponent range: nothing overflows at 2^16), the
    logits come back in the autocast dtype, and an inf gradient makes the scaler skip the step and back off."""
    spec, z, sd, x, meta, drops = load_case("tiny_b", golden_dir)
    xs, ms = x.cuda(), meta.cuda()
    tg = {t: torch.randint(0, c, (x.shape[0],), generator=torch.Generator().manual_seed(5)).cuda() for t, c in spec.heads}

    def one_step(use_scaler):
        model = build("tiny_b", spec, sd, dtype)
        model.train(True)
        model._inject_drop = drops
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16, enabled=use_scaler)
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_scaler):
            out = model(xs, ms)
            if use_scaler:
                assert all(v.dtype == torch.float16 for v in out.values())
            loss = sum(torch.nn.functional.cross_entropy(out[t].float(), tg[t]) for t, _ in spec.heads)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        grads = [p_.grad.detach().clone() for p_ in model.parameters()]
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        before = [p_.detach().clone() for p_ in model.parameters()]
        scaler.step(opt)
        scaler.update()
        assert any(not torch.equal(p_, q) for p_, q in zip(model.parameters(), before)), "the optimizer step did not happen"
        return model, opt, scaler, float(norm), float(loss.detach()), grads

    m0, _, _, n0, l0, g0 = one_step(False)
    m1, opt1, sc1, n1, l1, g1 = one_step(True)
    # fp16 logits feed the loss in the scaled flow: loss and (unscaled) gradients differ by that rounding only
    assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)) and abs(n0 - n1) <= 5e-3 * max(n0, 1e-3), (l0, l1, n0, n1)
    err = sum((a - b).double().pow(2).sum().item() for a, b in zip(g0, g1)) ** 0.5
    ref = sum(a.double().pow(2).sum().item() for a in g0) ** 0.5
    assert err <= 5e-3 * ref, (err, ref)
    # an overflowing gradient: the scaler must see the inf in the model's .grad views, skip the step and halve the scale
    before = [p_.detach().clone() for p_ in m1.parameters()]
    with torch.autocast("cuda", dtype=torch.float16):
        out = m1(xs, ms)
        loss = sum(torch.nn.functional.cross_entropy(out[t].float(), tg[t]) for t, _ in spec.heads)
    sc1.scale(loss).backward()
    next(iter(m1.parameters())).grad.view(-1)[0] = float("inf")
    sc1.unscale_(opt1)
    s_before = sc1.get_scale()
    sc1.step(opt1)
    sc1.update()
    assert sc1.get_scale() == s_before * 0.5
    for p_, q in zip(m1.parameters(), before):
        assert torch.equal(p_, q)


def test_hierarchical_softmax_heads_match_oracle(golden_dir):
    """A15: MODEL.CLASSIFICATION.HEADS of TYPE HierarchicalSoftmax (heads/hierarchical_softmax_head.py:28-210) -- effectively
    one shared Linear per task (finding F3), so logits and gradients are the tiny_c fixture's."""
    spec, z, sd, x, meta, drops = load_case("tiny_c", golden_dir)
    tree = TinyTree({"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}},
                    [t for t, _ in spec.heads], {t: c for t, c in spec.heads})
    model = build_model(make_config(spec, IMG["tiny_c"], "HierarchicalSoftmax"), num_classes={t: c for t, c in spec.heads}, taxonomy_tree=tree)
    assert all(type(h).__name__ == "HierarchicalSoftmaxHead" for h in model.head.values())
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype("fp32")
    out = run(model, x, meta, drops, train=True)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta, drops)
    for t, _ in spec.heads:
        torch.testing.assert_close(out[t].float().cpu(), oout[t].detach(), rtol=1e-4, atol=1e-4)
        assert (out[t].argmax(-1).cpu() == oout[t].argmax(-1)).all()
    O.probe_loss(out).backward()
    O.probe_loss(oout).backward()
    glob, wk = _grad_errors(model, osd)
    assert glob <= 1e-3, (glob, wk)


def test_device_prefetcher_delivers_batches_in_order():
    """linnaeus_amd.prefetch.DevicePrefetcher: tuples / dicts of host tensors (pinned or not) arrive on the GPU unchanged and
    in order, with `depth` transfers in flight, over several passes; device tensors pass through."""
    from linnaeus_amd.prefetch import DevicePrefetcher

    gen = torch.Generator().manual_seed(0)
    batches = []
    for i in range(7):
        img = torch.rand(4, 3, 32, 32, generator=gen)
        if i % 2 == 0:
            img = img.pin_memory()
        batches.append((img, {"taxa_L10": torch.randint(0, 9, (4,), generator=gen), "n": i}, [torch.full((2,), float(i))], torch.ones(1, device="cuda") * i))
    for depth in (1, 2, 3):
        pf = DevicePrefetcher(batches, depth=depth)
        assert len(pf) == 7
        for _ in range(2):
            seen = 0
            for i, (img, tgt, lst, dev_t) in enumerate(pf):
                assert img.is_cuda and tgt["taxa_L10"].is_cuda and lst[0].is_cuda and tgt["n"] == i
                # consume on the current stream right away (what a training step does)
                s = (img * 2).sum()
                assert torch.equal(img.cpu(), batches[i][0]) and torch.equal(tgt["taxa_L10"].cpu(), batches[i][1]["taxa_L10"])
                assert float(lst[0][0]) == float(i) and float(dev_t) == float(i)
                assert abs(float(s) - 2 * float(batches[i][0].sum())) < 1e-2
                seen += 1
            assert seen == 7
    with pytest.raises(ValueError):
        DevicePrefetcher(batches, depth=0)


def test_dropout_configs(golden_dir):
    """MODEL.DROP_RATE / ATTN_DROP_RATE > 0 (blocks/mlp.py:61-66, rope_2d_mhsa.py:497,503).  Dropout is the identity in eval
    mode, so such a model evaluates exactly like the same weights without dropout, and training forwards draw fresh masks
    (parity with injected masks: next test)."""
    spec, z, sd, x, meta, drops = load_case("tiny_a", golden_dir)
    ref = build("tiny_a", spec, sd, "fp32")
    cfg = make_config(spec, IMG["tiny_a"])
    cfg.MODEL.DROP_RATE = 0.1
    cfg.MODEL.ATTN_DROP_RATE = 0.05
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype("fp32")
    xs, ms = x.cuda(), meta.cuda() if meta is not None else None
    model.eval()
    ref.eval()
    with torch.no_grad():
        a, b = model(xs, ms), ref(xs, ms)
    for t in a:
        assert torch.equal(a[t], b[t]), t
    model.train()
    t1, t2 = model(xs, ms), model(xs, ms)
    assert any(not torch.equal(t1[t], t2[t]) for t in t1)
    assert any(not torch.equal(t1[t], a[t]) for t in t1)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_drop_rate_training_matches_reference(dtype, golden_dir):
    """the dropout path pinned to the reference.  tiny_drop.npz holds the keep mask of every nn.Dropout
    call the reference model made in one training forward (DROP_RATE 0.2, ATTN_DROP_RATE 0.1: blocks/mlp.py:61-66,
    rope_2d_mhsa.py:497,503) and its logits / loss / gradients; the HIP plan gets the same masks (lnx_plan_set_dropout /
    lnx_plan_set_attn_dropout layout) and must reproduce them -- kept-activation and recompute plans."""
    from tests.cases import load_dropout_case, plan_dropout_buffers

    spec, z, sd, x, meta, masks, ps = load_dropout_case(golden_dir)
    cfg = make_config(spec, 64)
    cfg.MODEL.DROP_RATE = float(z["drop_rate"])
    cfg.MODEL.ATTN_DROP_RATE = float(z["attn_drop_rate"])
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    model._inject_dropout, model._inject_attn_dropout = plan_dropout_buffers(masks)
    names = [str(n) for n in z["grad_names"]]
    for ck in (False, True):
        model.zero_grad(set_to_none=True)
        model.train(True)
        out = model(x.cuda(), meta.cuda(), force_checkpointing=ck)
        for t, _ in spec.heads:
            ref = torch.from_numpy(z["logits_" + t])
            got = out[t].float().cpu()
            if dtype == "fp32":
                torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, ref.abs().max().item()), msg=t)
                assert (got.argmax(-1) == ref.argmax(-1)).all()
            else:
                assert (got - ref).abs().max().item() <= 0.05 * max(1.0, ref.abs().max().item()), t
        loss = O.probe_loss(out)
        assert abs(loss.item() - float(z["loss"])) <= (1e-4 if dtype == "fp32" else 3e-2) * max(1.0, abs(float(z["loss"])))
        loss.backward()
        got = {}
        for k, p_ in model.named_parameters():
            parts = k.split(".")
            got[f"head.{parts[3]}.fc.{parts[4]}" if (parts[0] == "head" and len(parts) >= 5 and parts[2] == "level_classifiers") else k] = p_.grad
        num = den = 0.0
        for i, k in enumerate(names):
            ref_norm = float(z["grad_norms"][i])
            n = got[k].double().norm().item()
            num += (n - ref_norm) ** 2
            den += ref_norm ** 2
            if dtype == "fp32":
                assert abs(n - ref_norm) <= 1e-3 * max(ref_norm, 1e-3), (ck, k, n, ref_norm)
                np.testing.assert_allclose(got[k].reshape(-1)[:8].float().cpu().numpy(), z["gradslice_" + k], rtol=5e-3, atol=5e-6, err_msg=k)
        assert (num / den) ** 0.5 <= (1e-3 if dtype == "fp32" else 5e-2), (ck, (num / den) ** 0.5)


@pytest.mark.parametrize("name,dtype,attn", [("tiny_b", "fp32", False), ("tiny_dp", "fp32", True), ("tiny_b", "bf16", True), ("tiny_a", "fp32", True)])
def test_drop_rate_training_matches_oracle(name, dtype, attn, golden_dir):
    """MODEL.DROP_RATE = 0.2 in training: the two Mlp dropouts and proj_drop of every RoPE block, with the keep masks
    injected so that the CPU oracle applies the same ones; logits and gradients as in the dropout-free tests (the recompute
    plan included: it must replay the same masks).  `attn`: MODEL.ATTN_DROP_RATE = 0.1 on top -- dropout on the attention
    probabilities (rope_2d_mhsa.py:497), which runs the 64-row tiled attention kernels with the mask code.  A fresh draw
    (no injection) changes the output from call to call."""
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    cfg = make_config(spec, IMG[name])
    cfg.MODEL.DROP_RATE = 0.2
    cfg.MODEL.ATTN_DROP_RATE = 0.1 if attn else 0.0
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    model.load_state_dict(model_state_dict_from_oracle(model, sd), strict=True)
    model = model.cuda()
    model.set_compute_dtype(dtype)
    B = x.shape[0]
    E = 1 + len(spec.meta)
    side = IMG[name] // 16
    gen = torch.Generator().manual_seed(11)
    bufs, abufs, mult = [], [], []
    for s in range(2):
        N = (side >> s) ** 2 + E
        Np = (N + 63) // 64 * 64
        C, hid, h = spec.rope_dims[s], int(spec.rope_dims[s] * spec.mlp_ratio[s]), spec.rope_heads[s]
        for _ in range(spec.rope_depths[s]):
            trip = []
            for w in (C, hid, C):
                m = (torch.rand(B * N, w, generator=gen) < 0.8).to(torch.uint8)
                bufs.append(m.reshape(-1))
                trip.append(m.float().reshape(B, N, w) / 0.8)
            if attn:
                am = (torch.rand(B, h, N, Np, generator=gen) < 0.9).to(torch.uint8)
                abufs.append(am.reshape(-1))
                trip.append(am[..., :N].float() / 0.9)
            mult.append(trip)
    model._inject_dropout = torch.cat(bufs)
    if attn:
        model._inject_attn_dropout = torch.cat(abufs)
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oout = O.forward(osd, spec, x, meta, drops, dropout=mult)
    O.probe_loss(oout).backward()
    for ck in (False, True):
        model.zero_grad(set_to_none=True)
        model.train(True)
        model._inject_drop = drops
        out = model(x.cuda(), meta.cuda() if meta is not None else None, force_checkpointing=ck)
        for t, _ in spec.heads:
            ref = oout[t].detach()
            got = out[t].float().cpu()
            if dtype == "fp32":
                torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4 * max(1.0, ref.abs().max().item()), msg=t)
            else:
                assert (got - ref).abs().max().item() <= 0.05 * max(1.0, ref.abs().max().item()), t
        O.probe_loss(out).backward()
        glob, wk = _grad_errors(model, osd)
        assert glob <= (1e-3 if dtype == "fp32" else 6e-2), (ck, glob, wk)
    # without injection every training forward draws new masks
    model._inject_dropout = model._inject_attn_dropout = None
    a = model(x.cuda(), meta.cuda() if meta is not None else None)
    b = model(x.cuda(), meta.cuda() if meta is not None else None)
    assert any(not torch.equal(a[t], b[t]) for t in a)
    model.eval()
    with torch.no_grad():
        c1 = model(x.cuda(), meta.cuda() if meta is not None else None)
        c2 = model(x.cuda(), meta.cuda() if meta is not None else None)
    assert all(torch.equal(c1[t], c2[t]) for t in c1)
