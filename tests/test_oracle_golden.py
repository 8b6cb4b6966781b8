"""The oracle (oracle/mformer_oracle.py) against the reference's own outputs.

The fixtures under tests/golden/ were produced by tests/golden/gen/make_golden.py, which
imports the reference in the build container.  These tests pin the CPU restatement; they
need neither a GPU nor /root/reference.
"""
import numpy as np
import pytest
import torch

from oracle import mformer_oracle as O
from tests.cases import CASES, load_case


def _checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().max().item(), t.mean().item(), (t * t).sum().sqrt().item()])


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_dp", "sm"])
def test_forward_matches_reference(name, golden_dir):
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    taps = {}
    with torch.no_grad():
        out = O.forward(sd, spec, x, meta, drops, tap=lambda n, v: taps.__setitem__(n, v))
    np.testing.assert_allclose(taps["feats"].numpy(), z["feats"], rtol=1e-4, atol=2e-5)
    for task, _ in spec.heads:
        ref = z["logits_" + task]
        np.testing.assert_allclose(out[task].numpy(), ref, rtol=1e-4, atol=2e-5)
        # class-index argmax must be exact against the reference
        assert (out[task].argmax(-1).numpy() == ref.argmax(-1)).all()
    for tname in ("stem", "stage0", "down0", "stage1", "down1", "rope0", "down2", "rope1"):
        ref = z["tap_" + tname]
        got = _checksum(taps[tname])
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-4, err_msg=tname)
        np.testing.assert_allclose(taps[tname].reshape(-1)[:16].numpy(), z["tapslice_" + tname], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["tiny_a", "tiny_b", "tiny_c", "tiny_dp"])
def test_backward_matches_reference(name, golden_dir):
    spec, z, sd, x, meta, drops = load_case(name, golden_dir)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(sd, spec, x, meta, drops)
    loss = O.probe_loss(out)
    assert abs(loss.item() - float(z["loss"])) < 1e-5 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    names = [str(n) for n in z["grad_names"]]
    assert sorted(k for k, v in sd.items() if v.grad is not None) == names
    for i, k in enumerate(names):
        g = sd[k].grad
        ref_norm = z["grad_norms"][i]
        assert abs(g.double().norm().item() - ref_norm) <= 2e-4 * max(ref_norm, 1e-3), k
        np.testing.assert_allclose(g.reshape(-1)[:8].numpy(), z["gradslice_" + k], rtol=2e-3, atol=2e-6, err_msg=k)


def test_dropout_training_matches_reference(golden_dir):
    """MODEL.DROP_RATE = 0.2 / ATTN_DROP_RATE = 0.1 in training (blocks/mlp.py:61-66, rope_2d_mhsa.py:497,503): the oracle with
    the keep masks the reference drew (tiny_drop.npz) against the reference's logits, loss and gradients."""
    from tests.cases import load_dropout_case

    spec, z, sd, x, meta, masks, ps = load_dropout_case(golden_dir)
    assert ps == [0.1, 0.2, 0.2, 0.2] * sum(spec.rope_depths)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.forward(sd, spec, x, meta, None, dropout=O.dropout_multipliers_from_masks(masks, ps))
    for task, _ in spec.heads:
        np.testing.assert_allclose(out[task].detach().numpy(), z["logits_" + task], rtol=1e-4, atol=2e-5)
    loss = O.probe_loss(out)
    assert abs(loss.item() - float(z["loss"])) < 1e-5 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    names = [str(n) for n in z["grad_names"]]
    for i, k in enumerate(names):
        ref_norm = z["grad_norms"][i]
        assert abs(sd[k].grad.double().norm().item() - ref_norm) <= 2e-4 * max(ref_norm, 1e-3), k
        np.testing.assert_allclose(sd[k].grad.reshape(-1)[:8].numpy(), z["gradslice_" + k], rtol=2e-3, atol=2e-6, err_msg=k)
    # the masks matter: without them the logits differ
    with torch.no_grad():
        plain = O.forward(sd, spec, x, meta, None)
    assert any(np.abs(plain[t].numpy() - z["logits_" + t]).max() > 1e-3 for t, _ in spec.heads)


def test_sm_backward_grad_norms(golden_dir):
    spec, z, sd, x, meta, drops = load_case("sm", golden_dir)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = O.probe_loss(O.forward(sd, spec, x, meta, drops))
    loss.backward()
    names = [str(n) for n in z["grad_names"]]
    for i, k in enumerate(names):
        ref_norm = z["grad_norms"][i]
        assert abs(sd[k].grad.double().norm().item() - ref_norm) <= 5e-4 * max(ref_norm, 1e-3), k


def test_param_inventory_sm():
    shapes = O.param_shapes(CASES["sm"])
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == 30_265_691  # sm with four Linear heads 1000/300/80/20
    n_backbone = sum(int(np.prod(s)) for k, s in shapes.items() if not k.startswith("head."))
    assert n_backbone == 29_189_091 and len([k for k in shapes if not k.startswith("head.")]) == 225


def test_per_op_known_answers(golden_dir):
    z = np.load(f"{golden_dir}/per_op.npz")
    T = lambda k: torch.from_numpy(z[k])  # noqa: E731
    S = 20251003
    # F1: the "rotation" is a cos-only scaling
    cos = O.rope_cos_table(T("rope_freqs"), 3, 5)
    np.testing.assert_allclose(cos.numpy(), z["rope_cos_3x5"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.rope_scale_pairs(T("rope_q"), cos).numpy(), z["rope_q_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(O.rope_scale_pairs(T("rope_k"), cos).numpy(), z["rope_k_out"], rtol=1e-6, atol=1e-6)
    y = O.layer_norm_channels_first(T("lncf_x"), T("lncf_w"), T("lncf_b"), 1e-6)
    np.testing.assert_allclose(y.numpy(), z["lncf_y"], rtol=1e-5, atol=1e-6)
    # channels-first LN == channels-last LN on the permuted tensor
    y2 = O.layer_norm_last(T("lncf_x").permute(0, 2, 3, 1), T("lncf_w"), T("lncf_b"), 1e-6).permute(0, 3, 1, 2)
    np.testing.assert_allclose(y2.numpy(), z["lncf_y"], rtol=1e-5, atol=1e-6)
    shp = {"gamma": (8,), "dwconv.weight": (8, 1, 7, 7), "dwconv.bias": (8,), "norm.weight": (8,), "norm.bias": (8,),
           "pwconv1.weight": (32, 8), "pwconv1.bias": (32,), "pwconv2.weight": (8, 32), "pwconv2.bias": (8,)}
    sd = {"stages.0.0." + k: O.seeded_fill("stages.0.0." + k, s, S) for k, s in shp.items()}
    np.testing.assert_allclose(
        O.depthwise_conv7(T("cnb_x"), sd["stages.0.0.dwconv.weight"], sd["stages.0.0.dwconv.bias"]).numpy(), z["cnb_dw"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.convnext_block(sd, "stages.0.0.", T("cnb_x"), None).numpy(), z["cnb_y"], rtol=1e-5, atol=2e-6)
    shp = {"norm.weight": (8,), "norm.bias": (8,), "conv.weight": (16, 8, 2, 2), "conv.bias": (16,)}
    sd = {"downsample_layers.0." + k: O.seeded_fill("downsample_layers.0." + k, s, S) for k, s in shp.items()}
    np.testing.assert_allclose(O.downsample(sd, "downsample_layers.0.", T("ds_x")).numpy(), z["ds_y"], rtol=1e-5, atol=2e-6)
    shp = {"norm1.weight": (128,), "norm1.bias": (128,), "norm2.weight": (128,), "norm2.bias": (128,), "attn.freqs": (2, 2, 32),
           "attn.qkv.weight": (384, 128), "attn.qkv.bias": (384,), "attn.proj.weight": (128, 128), "attn.proj.bias": (128,),
           "mlp.fc1.weight": (512, 128), "mlp.fc1.bias": (512,), "mlp.fc2.weight": (128, 512), "mlp.fc2.bias": (128,)}
    sd = {"stages.2.0." + k: O.seeded_fill("stages.2.0." + k, s, S) for k, s in shp.items()}
    np.testing.assert_allclose(O.rope_attention(sd, "stages.2.0.attn.", T("att_x"), 3, 5, 2, 3).numpy(), z["att_y"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(O.rope_block(sd, "stages.2.0.", T("att_x"), 3, 5, 2, 3, None, None).numpy(), z["rb_y"], rtol=1e-5, atol=3e-6)
    shp = {"0.weight": (32, 3), "0.bias": (32,), "2.weight": (32,), "2.bias": (32,), "3.norm_fn1.weight": (32,), "3.norm_fn1.bias": (32,),
           "3.norm_fn2.weight": (32,), "3.norm_fn2.bias": (32,), "3.w1.weight": (32, 32), "3.w1.bias": (32,), "3.w2.weight": (32, 32), "3.w2.bias": (32,)}
    sd = {"meta_spatial_head_1." + k: O.seeded_fill("meta_spatial_head_1." + k, s, S) for k, s in shp.items()}
    np.testing.assert_allclose(O.meta_head(sd, "meta_spatial_head_1.", T("mh_x")).numpy(), z["mh_y"], rtol=1e-5, atol=2e-6)


def test_train_step_matches_reference(golden_dir):
    """Caller (ii): loss, pre-clip gradient norm and parameter deltas of two optimizer steps."""
    from tests.cases import load_train_step, train_steps

    spec, z, sd, x, meta, targets, weights = load_train_step(golden_dir)
    before = {k: v.clone() for k, v in sd.items()}
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    losses, norms = train_steps(lambda: O.forward(sd, spec, x, meta), list(sd.values()), z, spec, targets, weights)
    for s in range(int(z["steps"])):
        assert abs(losses[s] - float(z[f"loss_{s}"])) <= 2e-5 * abs(float(z[f"loss_{s}"])), (s, losses[s])
        assert abs(norms[s] - float(z[f"gnorm_{s}"])) <= 2e-4 * float(z[f"gnorm_{s}"]), (s, norms[s])
    names = [str(n) for n in z["param_names"]]
    assert sorted(sd) == names
    for i, k in enumerate(names):
        d = (sd[k].detach() - before[k]).double().norm().item()
        if k == "aggregate.bias":
            continue  # a constant shift in front of final_norm: its gradient is exactly zero in real arithmetic, and Adam turns the rounding noise into +-lr
        floor = 0.05 * float(z["lr"]) * int(z["steps"]) * sd[k].numel() ** 0.5
        assert abs(d - z["delta_norms"][i]) <= 2e-2 * z["delta_norms"][i] + floor, (k, d, z["delta_norms"][i])


HIER_MODES = {"sched1": (1.0, False, False, True), "sched0": (0.0, False, False, True), "phase1": (1.0, True, False, True), "val": (0.0, False, True, True),
              "sched0_nocw": (0.0, False, False, False)}


def _hier_fixture(golden_dir):
    z = np.load(f"{golden_dir}/hier_loss.npz")
    tasks = [str(t) for t in z["tasks"]]
    return z, tasks, [int(c) for c in z["classes"]]


def test_oracle_smoothing_matrix_matches_reference(golden_dir):
    z, tasks, classes = _hier_fixture(golden_dir)
    for t, c in zip(tasks, classes):
        for beta, ur in ((1.0, 1), (0.5, 0)):
            m = O.taxonomy_smoothing_matrix(c, torch.from_numpy(z[f"dist_{t}"]), alpha=0.15, beta=beta, uniform_roots=bool(ur), root_class_ids=list(z[f"roots_{t}"]))
            np.testing.assert_allclose(m.numpy(), z[f"smooth_{t}_b{beta}_u{ur}"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(m.sum(1).numpy(), 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("mode", list(HIER_MODES))
def test_oracle_hierarchical_loss_matches_reference(mode, golden_dir):
    """weighted_hierarchical_loss of the reference (fixture made by tests/golden/gen/make_golden.py hier_loss) vs the
    oracle restatement: total, per-task weighted losses and the logits gradients of every masking mode (F13 included)."""
    z, tasks, classes = _hier_fixture(golden_dir)
    prob, phase1, val, use_cw = HIER_MODES[mode]
    lg = {t: torch.from_numpy(z[f"logits_{t}"]).clone().requires_grad_(True) for t in tasks}
    tg = {t: torch.from_numpy(z[f"target_{t}"]) for t in tasks}
    soft = {t: torch.from_numpy(z[f"soft_{t}"]) for t in tasks}
    cw = {t: torch.from_numpy(z[f"cw_{t}"]) for t in tasks} if use_cw else None
    tw = {t: float(w) for t, w in zip(tasks, z["task_weights"])}
    total, weighted = O.hierarchical_loss(lg, tg, soft, tw, cw, prob, phase1, val)
    total.backward()
    assert abs(total.item() - float(z[f"{mode}_total"])) <= 2e-5 * abs(float(z[f"{mode}_total"]))
    np.testing.assert_allclose([weighted[t].item() for t in tasks], z[f"{mode}_weighted"], rtol=2e-5)
    for t in tasks:
        np.testing.assert_allclose(lg[t].grad.numpy(), z[f"{mode}_grad_{t}"], rtol=1e-4, atol=1e-6)


def test_oracle_newton_schulz_matches_reference(golden_dir):
    """Muon's orthogonalisation (optimizers/muon.py:27-65): the oracle's bf16 restatement reproduces the reference bit for bit"""
    z = np.load(f"{golden_dir}/muon.npz")
    for name in ("wide", "tall", "odd", "sq"):
        got = O.newton_schulz5(torch.from_numpy(z[f"ns_in_{name}"]), steps=5).float().numpy()
        np.testing.assert_array_equal(got, z[f"ns_out_{name}"])
