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
