"""GPU-side batch mixing: the oracle restatement against the reference's fixture (CPU), and the HIP kernels
behind linnaeus_amd.collate against the same fixture with the reference's recorded random draws (GPU)."""
import numpy as np
import pytest
import torch

from oracle import mformer_oracle as O


def _load(golden_dir):
    z = np.load(f"{golden_dir}/collate.npz")
    t = {k: torch.from_numpy(z[k]) for k in ("images", "aux", "masks", "gids")}
    targets = {k[len("target_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("target_")}
    bounds = [tuple(int(v) for v in r) for r in z["bounds"]]
    return z, t, targets, bounds


def _check(z, name, out, tol=0.0):
    mi, mt, ma, mm = out
    np.testing.assert_allclose(mi.cpu().numpy(), z[f"{name}_images"], rtol=tol, atol=tol)
    for k, v in mt.items():
        np.testing.assert_allclose(v.cpu().numpy(), z[f"{name}_target_{k}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(ma.cpu().numpy(), z[f"{name}_aux"], rtol=0, atol=0)
    assert np.array_equal(mm.cpu().numpy().astype(bool), z[f"{name}_masks"])


def test_oracle_mixup_cutmix_match_reference(golden_dir):
    z, t, targets, bounds = _load(golden_dir)
    out = O.selective_mixup(t["images"], targets, t["aux"], t["masks"], t["gids"], torch.from_numpy(z["mixup_perm"]), float(z["mixup_lam"]),
                            torch.from_numpy(z["mixup_pick"]), bounds)
    _check(z, "mixup", out, 1e-6)
    out = O.selective_cutmix(t["images"], targets, t["aux"], t["masks"], t["gids"], torch.from_numpy(z["cutmix_perm"]), z["cutmix_box"],
                             torch.from_numpy(z["cutmix_pick"]), bounds)
    _check(z, "cutmix", out, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixup", "cutmix"])
def test_hip_collate_matches_reference(name, golden_dir):
    from linnaeus_amd.collate import GPUSelectiveCutMix, GPUSelectiveMixup

    z, t, targets, bounds = _load(golden_dir)
    cls = GPUSelectiveMixup if name == "mixup" else GPUSelectiveCutMix
    op = cls({"PROB": 1.0, "ALPHA": 0.4 if name == "mixup" else 1.0, "meta_chunk_bounds_list": [list(b) for b in bounds]})
    op._inject = {"perm": torch.from_numpy(z[f"{name}_perm"]), "lam": float(z[f"{name}_lam"]), "pick": torch.from_numpy(z[f"{name}_pick"]),
                  "box": tuple(int(v) for v in z["cutmix_box"]) if name == "cutmix" else None}
    batch = (t["images"].cuda(), {k: v.cuda() for k, v in targets.items()}, t["aux"].cuda(), t["masks"].cuda(), t["gids"].cuda())
    out = op(batch, exclude_null_samples=True)
    _check(z, name, out, 1e-6)
    assert torch.equal(batch[2].cpu(), t["aux"])  # the caller's metadata is not modified


@pytest.mark.gpu
def test_hip_collate_random_draws_and_shapes():
    """Without injected draws: partners stay inside their group, null / ungrouped samples are untouched, a mixed image is
    a convex combination of the pair, B = 256 x 3 x 224 x 224 runs, ragged class counts (C % 4 != 0) work."""
    from linnaeus_amd.collate import GPUSelectiveCutMix, GPUSelectiveMixup, ingroup_permutation

    g = torch.Generator().manual_seed(0)
    gids = torch.randint(-1, 5, (64,), generator=g).cuda()
    for _ in range(5):
        perm = ingroup_permutation(gids)
        assert torch.equal(torch.sort(perm).values, torch.arange(64, device="cuda"))
        assert torch.equal(gids[perm], gids) and torch.equal(perm[gids == -1], torch.arange(64, device="cuda")[gids == -1])
    B = 256
    x = torch.rand(B, 3, 224, 224, device="cuda")
    lab = torch.randint(0, 7, (B,), device="cuda")
    tg = {"taxa_L10": torch.nn.functional.one_hot(lab, 7).float()}
    aux = torch.randn(B, 5, device="cuda")
    gid = torch.randint(0, 8, (B,), device="cuda")
    mix = GPUSelectiveMixup({"PROB": 1.0, "ALPHA": 0.4, "meta_chunk_bounds_list": [[0, 2], [2, 5]]})
    mi, mt, ma, mm = mix((x, tg, aux, aux != 0, gid))
    perm = mix.last_permutation
    null = lab == 0
    assert torch.equal(perm[null], torch.arange(B, device="cuda")[null])
    torch.testing.assert_close(mi[null], x[null], rtol=1e-6, atol=1e-7)  # lam x + (1 - lam) x, as the reference computes it
    lo, hi = torch.minimum(x, x[perm]), torch.maximum(x, x[perm])
    assert bool(((mi >= lo - 1e-6) & (mi <= hi + 1e-6)).all())
    torch.testing.assert_close(mt["taxa_L10"].sum(1), torch.ones(B, device="cuda"))
    cut = GPUSelectiveCutMix({"PROB": 1.0, "ALPHA": 1.0, "meta_chunk_bounds_list": [[0, 2], [2, 5]]})
    ci, ct, _, _ = cut((x, tg, aux, aux != 0, gid))
    same = (ci == x) | (ci == x[cut.last_permutation])
    assert bool(same.all())
