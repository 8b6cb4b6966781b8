"""Loss on device: the oracle restatement against the reference's fixture (CPU), and the HIP kernel
behind linnaeus_amd.loss against the same fixture and against torch's cross_entropy (GPU)."""
import numpy as np
import pytest
import torch

from oracle import mformer_oracle as O

VARIANTS = {"plain": {}, "ignore0": {"ignore_index": 0}, "ignore0_cw": {"ignore_index": 0, "cw": True}, "cw": {"cw": True}}


def _load(golden_dir):
    z = np.load(f"{golden_dir}/soft_ce.npz")
    return z, {k: torch.from_numpy(z[k]) for k in ("logits", "target", "soft", "class_weight", "wsum")}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_oracle_soft_ce_matches_reference(name, golden_dir):
    z, t = _load(golden_dir)
    v = VARIANTS[name]
    x = t["logits"].clone().requires_grad_(True)
    loss = O.soft_label_ce(x, t["target"], t["soft"], t["class_weight"] if v.get("cw") else None, v.get("ignore_index"))
    np.testing.assert_allclose(loss.detach().numpy(), z[f"loss_{name}"], rtol=1e-5, atol=1e-6)
    (loss * t["wsum"]).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), z[f"grad_{name}"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_hip_soft_ce_matches_reference(name, golden_dir):
    from linnaeus_amd.loss import TaxonomyAwareLabelSmoothingCE

    z, t = _load(golden_dir)
    v = VARIANTS[name]
    crit = TaxonomyAwareLabelSmoothingCE(t["soft"], weight=t["class_weight"] if v.get("cw") else None, apply_class_weights=bool(v.get("cw")),
                                         ignore_index=v.get("ignore_index")).cuda()
    x = t["logits"].cuda().requires_grad_(True)
    loss = crit(x, t["target"].cuda())
    np.testing.assert_allclose(loss.detach().cpu().numpy(), z[f"loss_{name}"], rtol=2e-5, atol=2e-6)
    (loss * t["wsum"].cuda()).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), z[f"grad_{name}"], rtol=2e-4, atol=2e-6)


@pytest.mark.gpu
def test_hip_soft_ce_interface_errors():
    from linnaeus_amd.loss import TaxonomyAwareLabelSmoothingCE

    with pytest.raises(ValueError):
        TaxonomyAwareLabelSmoothingCE(torch.rand(3, 4))
    crit = TaxonomyAwareLabelSmoothingCE(torch.eye(5)).cuda()
    with pytest.raises(ValueError):
        crit(torch.randn(2, 6).cuda(), torch.zeros(2, dtype=torch.long).cuda())
    with pytest.raises(IndexError):
        crit(torch.randn(2, 5).cuda(), torch.tensor([1, 7]).cuda())
    # dict input (ConditionalClassifierHead style) and one-hot targets are accepted like the reference
    lg = torch.randn(3, 5).cuda()
    a = crit({"aux": torch.randn(3, 2).cuda(), "logits": lg}, torch.nn.functional.one_hot(torch.tensor([0, 2, 4]), 5).cuda())
    torch.testing.assert_close(a, torch.nn.functional.cross_entropy(lg, torch.tensor([0, 2, 4]).cuda(), reduction="none"), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("smoothing", [0.0, 0.1])
def test_hip_multitask_cross_entropy_matches_torch(smoothing):
    """the loss of the throughput protocol: sum_t w_t * mean_b CE, on padded-pitch logits views of several widths"""
    from linnaeus_amd.loss import multitask_cross_entropy

    g = torch.Generator().manual_seed(3)
    B = 64
    tasks = {"taxa_L10": 1000, "taxa_L20": 300, "taxa_L30": 80, "taxa_L40": 20}
    w = {"taxa_L10": 1.0, "taxa_L20": 0.5, "taxa_L30": 0.25, "taxa_L40": 2.0}
    outs, outs_ref, tg = {}, {}, {}
    for t, c in tasks.items():
        base = (torch.randn(B, c + 8, generator=g) * 2).cuda()  # logits as a column slice of a wider buffer
        outs[t] = base[:, :c].clone().requires_grad_(True) if c == 20 else base.requires_grad_(True)[:, :c]
        outs_ref[t] = base.detach()[:, :c].clone().requires_grad_(True)
        tg[t] = torch.randint(0, c, (B,), generator=g).cuda()
    loss = multitask_cross_entropy(outs, tg, w, label_smoothing=smoothing)
    ref = sum(w[t] * torch.nn.functional.cross_entropy(outs_ref[t], tg[t], label_smoothing=smoothing) for t in tasks)
    torch.testing.assert_close(loss, ref, rtol=1e-5, atol=1e-5)
    gl = torch.autograd.grad(loss * 3.0, list(outs.values()))
    gr = torch.autograd.grad(ref * 3.0, list(outs_ref.values()))
    for a, b in zip(gl, gr):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sched1", "sched0", "phase1", "val", "sched0_nocw"])
def test_hip_hierarchical_loss_matches_reference(mode, golden_dir):
    """linnaeus_amd.loss.weighted_hierarchical_loss (HIP soft-label criterion + device-side masking / class weights / task
    weights, same call signature as loss/hierarchical_loss.py:24) against the numbers the reference produced."""
    from types import SimpleNamespace as NS

    from linnaeus_amd.loss import GradientWeighting, TaxonomyAwareLabelSmoothingCE, build_taxonomy_smoothing_matrix, weighted_hierarchical_loss
    from tests.test_oracle_golden import HIER_MODES, _hier_fixture

    z, tasks, classes = _hier_fixture(golden_dir)
    prob, phase1, val, use_cw = HIER_MODES[mode]
    cfg = NS(TRAIN=NS(PHASE1_MASK_NULL_LOSS=phase1), LOSS=NS(GRAD_WEIGHTING=NS(CLASS=NS(TRAIN=True, VAL=False))))
    lg = {t: torch.from_numpy(z[f"logits_{t}"]).cuda().requires_grad_(True) for t in tasks}
    tg = {t: torch.from_numpy(z[f"target_{t}"]).cuda() for t in tasks}
    crit = {}
    for t, c in zip(tasks, classes):
        m = build_taxonomy_smoothing_matrix(c, torch.from_numpy(z[f"dist_{t}"]).cuda(), alpha=0.15, beta=1.0, uniform_roots=True, root_class_ids=list(z[f"roots_{t}"]))
        np.testing.assert_allclose(m.cpu().numpy(), z[f"soft_{t}"], rtol=1e-6, atol=1e-7)
        crit[t] = TaxonomyAwareLabelSmoothingCE(m).cuda()
        crit[t].validate_targets = False  # no host sync in the step
    cw = {t: {i: float(z[f"cw_{t}"][i]) for i in range(0, c, 2)} for t, c in zip(tasks, classes)} if use_cw else None
    gw = GradientWeighting(tasks, cfg, "static", init_weights={t: float(w) for t, w in zip(tasks, z["task_weights"])}, class_weights=cw)
    sched = NS(get_null_mask_prob=lambda step: prob)
    total, comps, weights = weighted_hierarchical_loss(lg, tg, crit, gw, sched, 10, is_validation=val, config=cfg)
    total.backward()
    assert abs(total.item() - float(z[f"{mode}_total"])) <= 2e-5 * abs(float(z[f"{mode}_total"]))
    np.testing.assert_allclose([comps["weighted_tasks"][t] for t in tasks], z[f"{mode}_weighted"], rtol=2e-5)
    np.testing.assert_allclose([comps["tasks"][t] for t in tasks], z[f"{mode}_raw_mean"], rtol=2e-5)
    np.testing.assert_allclose([comps["masked_tasks"][t] for t in tasks], z[f"{mode}_masked_mean"], rtol=2e-5)
    for t in tasks:
        np.testing.assert_allclose(lg[t].grad.cpu().numpy(), z[f"{mode}_grad_{t}"], rtol=2e-4, atol=2e-6)
    assert weights == pytest.approx({t: float(w) for t, w in zip(tasks, z["task_weights"])})


@pytest.mark.gpu
def test_hip_hierarchical_loss_scheduled_fraction():
    """Fractional inclusion probability: the kept set follows the injected uniform draws (the reference draws its own)."""
    from types import SimpleNamespace as NS

    from linnaeus_amd.loss import GradientWeighting, TaxonomyAwareLabelSmoothingCE, weighted_hierarchical_loss

    B, Cn = 12, 5
    g = torch.Generator().manual_seed(1)
    lg = {"taxa_L10": torch.randn(B, Cn, generator=g).cuda().requires_grad_(True)}
    tg = {"taxa_L10": torch.tensor([0, 0, 0, 0, 1, 2, 3, 4, 1, 2, 0, 0]).cuda()}
    crit = {"taxa_L10": TaxonomyAwareLabelSmoothingCE(torch.eye(Cn)).cuda()}
    coin = {"taxa_L10": torch.tensor([0.1, 0.9, 0.4, 0.6, 0, 0, 0, 0, 0, 0, 0.49, 0.51])}
    cfg = NS(TRAIN=NS(PHASE1_MASK_NULL_LOSS=False), LOSS=NS(GRAD_WEIGHTING=NS(CLASS=NS(TRAIN=True, VAL=False))))
    gw = GradientWeighting(["taxa_L10"], cfg, "static")
    total, comps, _ = weighted_hierarchical_loss(lg, tg, crit, gw, NS(get_null_mask_prob=lambda s: 0.5), 0, config=cfg, _coin=coin)
    per = torch.nn.functional.cross_entropy(lg["taxa_L10"].detach(), tg["taxa_L10"], reduction="none")
    keep = torch.tensor([1, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0], dtype=torch.bool).cuda()
    torch.testing.assert_close(total.detach(), (per * keep).sum() / keep.sum(), rtol=1e-5, atol=1e-6)
    assert int(comps["null_masking"]["null_samples_total"]) == 6 and int(comps["null_masking"]["null_samples_included"]) == 3
