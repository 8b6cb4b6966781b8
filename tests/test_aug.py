"""SURVEY 8f-3 remainder: GPU image augmentations (linnaeus_amd/aug.py over csrc/aug.hip) and the flat-file synthetic reader.

tests/golden/aug.npz holds outputs of the reference's own GPUAutoAugmentBatch / GPURandomErasing (CPU tensors, build
container) for everything of them that runs; the operations that raise upstream are checked against the torch restatements of
oracle/aug_oracle.py ("parity unpinned", see its header)."""
import os

import numpy as np
import pytest
import torch

from oracle import aug_oracle as A

gpu = pytest.mark.gpu
RE_MODES = ("const", "rand", "pixel")


def _z(golden_dir):
    return np.load(f"{golden_dir}/aug.npz", allow_pickle=False)


def _re_cfg(mode):
    return {"PROB": 0.9, "AREA_RANGE": [0.02, 0.3], "ASPECT_RATIO": [0.3, 3.3], "COUNT": 2, "MODE": mode}


def _re_draws(z, mode, seed):
    draws = []
    for it in range(2):
        d = {k: torch.from_numpy(z[f"re_{mode}_{seed}_{it}_{k}"]) for k in ("areas", "aspects", "x", "y", "values")}
        if it == 0:
            d["gate"] = torch.from_numpy(z[f"re_{mode}_{seed}_gate"])
        draws.append(d)
    return draws


# ---------------------------------------------------------------- CPU: oracle and host logic
def test_oracle_pointwise_ops_match_reference(golden_dir):
    z = _z(golden_dir)
    img = torch.from_numpy(z["img"])
    fns = {"Posterize": lambda m: A.posterize(img, m), "PosterizeOriginal": lambda m: A.posterize(img, m), "PosterizeIncreasing": lambda m: A.posterize(img, 8 - m),
           "Solarize": lambda m: A.solarize(img, m), "SolarizeAdd": lambda m: A.solarize_add(img, m), "Invert": lambda m: A.invert(img)}
    for key in z["working_ops"]:
        op, m = str(key).split(":")
        assert torch.equal(fns[op](int(m) * 0.1), torch.from_numpy(z[f"op_{op}_{m}"])), key
    assert len(z["raising_ops"]) == 14  # what the reference cannot run (recorded when the fixture was made)


@pytest.mark.parametrize("mode", RE_MODES)
def test_oracle_random_erasing_matches_reference(mode, golden_dir):
    z = _z(golden_dir)
    img = torch.from_numpy(z["re_img"])
    for seed in (11, 12):
        out = A.random_erasing(img, _re_draws(z, mode, seed), _re_cfg(mode))
        assert torch.equal(out, torch.from_numpy(z[f"re_{mode}_{seed}_out"]))


def test_policies_and_errors():
    from linnaeus_amd.aug import get_policy

    for name, n in (("original", 25), ("originalr", 25), ("v0r", 25), ("3a", 3), ("hybrid_v0", 28)):
        pol = get_policy(name, {})
        assert len(pol) == n and all(0.0 <= p_ <= 1.0 and 0 <= m <= 10 for sub in pol for _, p_, m in sub)
    assert get_policy("original", {})[0] == [("PosterizeOriginal", 0.4, 8), ("Rotate", 0.6, 9)]
    assert get_policy("originalr", {})[0][0][0] == "PosterizeIncreasing"
    with pytest.raises(ValueError, match="Unknown AutoAugment policy"):
        get_policy("nope", {})
    if os.path.isdir("/root/reference"):  # build container: the tables are the reference's, entry by entry
        import subprocess
        import sys

        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import logging; logging.disable(logging.CRITICAL)\n"
                "from linnaeus.aug.policies import get_policy as ref\nfrom linnaeus_amd.aug import get_policy as ours\n"
                "for n in ('original', 'originalr', 'v0r', '3a', 'hybrid_v0'):\n    assert [[tuple(o) for o in s] for s in ref(n, {})] == ours(n, {}), n\nprint('OK')")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=os.pathsep.join([os.path.join(repo, "tests", "golden", "gen", "_stubs"), "/root/reference", repo]))
        r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_cpu_images_fail_loudly():
    from linnaeus_amd import _lib
    from linnaeus_amd.aug import GPUAutoAugmentBatch, GPURandomErasing

    with pytest.raises(_lib.LnxError, match="no CPU fallback"):
        GPUAutoAugmentBatch("3a", 0.4)(torch.rand(1, 3, 8, 8))
    with pytest.raises(_lib.LnxError, match="no CPU fallback"):
        GPURandomErasing(_re_cfg("const"))(torch.rand(1, 3, 8, 8))


def test_flat_dataset_read_contract(tmp_path):
    """prefetching_h5_dataset.py:185-360: (image [3, S, S] in [0, 1], one-hot targets with null -> index 0, aux_info in IDX order
    with null components zeroed, group id, subset ids, per-element validity mask); batches in collate_fn's layout."""
    from linnaeus_amd.flatdata import FlatBatchLoader, FlatSyntheticDataset, write_synthetic_flat

    path = str(tmp_path / "syn.flat")
    tasks = {"taxa_L10": 9, "taxa_L20": 4}
    write_synthetic_flat(path, 64, 16, tasks, meta=(("TEMPORAL", 2), ("SPATIAL", 3)), seed=7, null_fraction=0.3)
    ds = FlatSyntheticDataset(path)
    assert len(ds) == 64 and ds.tasks == list(tasks)
    seen_null = seen_invalid = False
    for i in range(64):
        img, tg, aux, gid, subset, mask = ds._read_raw_item(i)
        assert img.shape == (3, 16, 16) and img.dtype == torch.float32 and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
        assert torch.equal(img, torch.from_numpy(np.array(ds.raw_image(i))).permute(2, 0, 1).float() / 255.0)
        for t, c in tasks.items():
            assert tg[t].shape == (c,) and float(tg[t].sum()) == 1.0
            seen_null |= bool(tg[t][0] == 1.0)
        assert aux.shape == (5,) and mask.shape == (5,) and mask.dtype == torch.bool and isinstance(gid, int) and subset == {}
        for lo, hi in ((0, 2), (2, 5)):
            assert bool(mask[lo:hi].all()) or (not bool(mask[lo:hi].any()) and bool((aux[lo:hi] == 0).all()))
            seen_invalid |= not bool(mask[lo:hi].any())
    assert seen_null and seen_invalid
    got = list(FlatBatchLoader(ds, 16, shuffle=True, seed=1))
    assert len(got) == 4
    img, tg, aux, masks, gids = got[0]
    assert img.shape == (16, 3, 16, 16) and aux.shape == (16, 5) and masks.dtype == torch.bool and gids.dtype == torch.int64
    # every sample of the epoch exactly once, and the collated rows equal the per-sample reads
    order = np.random.default_rng(1).permutation(64)
    for bi, (img, tg, aux, masks, gids) in enumerate(got):
        for j in range(16):
            s = ds._read_raw_item(int(order[bi * 16 + j]))
            assert torch.equal(img[j], s[0]) and torch.equal(aux[j], s[2]) and torch.equal(masks[j], s[5]) and int(gids[j]) == s[3]
            assert all(torch.equal(tg[t][j], s[1][t]) for t in tasks)
    raw = next(iter(FlatBatchLoader(ds, 8, raw_uint8=True)))[0]
    assert raw.dtype == torch.uint8 and raw.shape == (8, 16, 16, 3)


# ---------------------------------------------------------------- GPU: the HIP kernels through the reference's class names
@gpu
def test_pointwise_ops_bit_exact_against_reference(golden_dir):
    from linnaeus_amd.aug import GPUAutoAugmentBatch

    z = _z(golden_dir)
    img = torch.from_numpy(z["img"]).cuda()
    aa = GPUAutoAugmentBatch("v0r", 0.4)
    for key in z["working_ops"]:
        op, m = str(key).split(":")
        out = aa._apply_op(img.clone(), op, int(m))
        assert torch.equal(out.cpu(), torch.from_numpy(z[f"op_{op}_{m}"])), key


@gpu
def test_autoaugment_call_makes_the_reference_s_choices(golden_dir):
    """__call__ (autoaug.py:88-103): same CPU coin flips in the same order -> the same operations -> identical images."""
    from linnaeus_amd.aug import GPUAutoAugmentBatch

    z = _z(golden_dir)
    img = torch.from_numpy(z["img"])
    aa = GPUAutoAugmentBatch("v0r", 0.4)
    aa.policy = [[(o, float(p_), int(m)) for o, p_, m in (it.split(":") for it in str(sub).split())] for sub in z["call_policy"]]
    for seed in (3, 4, 5):
        torch.manual_seed(seed)
        out = aa((img * 1.3 - 0.1).cuda())
        assert torch.equal(out.cpu(), torch.from_numpy(z[f"call_seed{seed}"])), seed


@gpu
@pytest.mark.parametrize("mode", RE_MODES)
def test_random_erasing_against_reference(mode, golden_dir):
    from linnaeus_amd.aug import GPURandomErasing

    z = _z(golden_dir)
    img = torch.from_numpy(z["re_img"])
    for seed in (11, 12):
        re = GPURandomErasing(_re_cfg(mode))
        re._draws = _re_draws(z, mode, seed)
        out = re(img.cuda().clone()).cpu()
        ref = torch.from_numpy(z[f"re_{mode}_{seed}_out"])
        if mode == "pixel":  # mean / std are reductions: summation order differs on the device
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(out, ref)


@gpu
def test_random_erasing_batches():
    """B > 1 (where the reference raises): per sample one rectangle of one value per channel, inside the image, sized within
    AREA_RANGE; PROB = 0 leaves the batch untouched; no host synchronisation is needed to decide."""
    from linnaeus_amd.aug import GPURandomErasing

    torch.manual_seed(0)
    x = torch.rand(32, 3, 48, 40, device="cuda")
    assert torch.equal(GPURandomErasing({**_re_cfg("const"), "PROB": 0.0})(x.clone()), x)
    cfg = {**_re_cfg("const"), "PROB": 1.0, "COUNT": 1}
    y = GPURandomErasing(cfg)(x.clone())
    changed = 0
    for b in range(32):
        diff = (y[b] != x[b]).any(0)
        if not bool(diff.any()):
            continue
        changed += 1
        rows, cols = torch.where(diff.any(1))[0], torch.where(diff.any(0))[0]
        h, w = int(rows[-1] - rows[0]) + 1, int(cols[-1] - cols[0]) + 1
        box = y[b, :, rows[0]:rows[0] + h, cols[0]:cols[0] + w]
        assert bool((box == box[:, :1, :1]).all()), b            # one value per channel
        assert 0.5 * 0.02 * 48 * 40 <= h * w <= 1.5 * 0.3 * 48 * 40, (b, h, w)
    assert changed >= 24
    assert 0.0 <= float(y.min()) and float(y.max()) <= 1.0


@gpu
def test_unpinned_ops_match_their_definitions():
    """The operations the reference cannot run, against oracle/aug_oracle.py's restatements of what they name."""
    from linnaeus_amd.aug import GPUAutoAugmentBatch, inverse_affine_matrix

    g = torch.Generator().manual_seed(5)
    img = torch.rand(4, 3, 36, 28, generator=g)
    aa = GPUAutoAugmentBatch("original", 0.4)

    def run(op, m):
        return aa._apply_op(img.cuda().clone(), op, m).cpu()

    tol = dict(rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(run("Color", 4), A.saturation(img, 1.4), **tol)
    torch.testing.assert_close(run("Desaturate", 10), A.saturation(img, 0.0), **tol)
    torch.testing.assert_close(run("Brightness", 6), A.brightness(img, 1.6), **tol)
    torch.testing.assert_close(run("Contrast", 8), A.contrast(img, 1.8), **tol)
    torch.testing.assert_close(run("AutoContrast", 5), A.autocontrast(img), **tol)
    torch.testing.assert_close(run("Equalize", 5), A.equalize(img), **tol)
    torch.testing.assert_close(run("Sharpness", 7), A.sharpness(img, 0.7), **tol)
    torch.testing.assert_close(run("GaussianBlurRand", 10), A.gaussian_blur(img, 1.0), **tol)
    assert torch.equal(run("GaussianBlurRand", 0), img)
    # geometric ops: autoaug magnitudes are <= 1 degree / 1 pixel, so also exercise the kernel with large transforms
    for op, m, kw in (("ShearX", 9, dict(shear=(0.9, 0.0))), ("ShearY", 4, dict(shear=(0.0, 0.4))), ("TranslateX", 10, dict(translate=(1.0, 0.0))),
                      ("TranslateY", 10, dict(translate=(0.0, 1.0))), ("TranslateYRel", 9, dict(translate=(0.0, 0.9 * 28))), ("Rotate", 9, dict(angle=-0.9))):
        m6 = inverse_affine_matrix(kw.get("angle", 0.0), kw.get("translate", (0.0, 0.0)), kw.get("shear", (0.0, 0.0)))
        got, want = run(op, m), A.affine(img, m6)
        assert (got != want).float().mean().item() < 2e-3, op  # nearest neighbour: only exact .5 ties may round differently
    for kw in (dict(angle=-33.0), dict(shear=(25.0, -10.0)), dict(angle=10.0, translate=(5.3, -7.1), shear=(8.0, 3.0))):
        m6 = inverse_affine_matrix(kw.get("angle", 0.0), kw.get("translate", (0.0, 0.0)), kw.get("shear", (0.0, 0.0)))
        got, want = aa._affine(img.cuda(), **kw).cpu(), A.affine(img, m6)
        assert (got != want).float().mean().item() < 5e-3, kw
        assert float((want == 0).float().mean()) > 0.02  # the zero fill is exercised
    with pytest.raises(ValueError, match="Unknown operation"):
        aa._apply_op(img.cuda(), "Nope", 1)


@gpu
def test_pipeline_and_uint8_path(tmp_path):
    """pipeline.py:59-103 on a single sample and on a batch; the raw uint8 batch of the flat reader through the prefetcher,
    converted on the device, equals the host-converted float batch."""
    from types import SimpleNamespace as NS

    from linnaeus_amd.aug import GPUAugmentationPipeline, u8hwc_to_f32chw
    from linnaeus_amd.flatdata import FlatBatchLoader, FlatSyntheticDataset, write_synthetic_flat
    from linnaeus_amd.prefetch import DevicePrefetcher

    cfg = NS(AUG=NS(AUTOAUG=NS(POLICY="hybrid_v0", COLOR_JITTER=0.4), RANDOM_ERASE={**_re_cfg("pixel"), "PROB": 1.0}))
    pipe = GPUAugmentationPipeline(cfg)
    torch.manual_seed(1)
    one = torch.rand(3, 32, 32, device="cuda") * 255.0  # 0..255 input is rescaled
    out, tg, aux = pipe((one, {"t": 1}, None))
    assert out.shape == (3, 32, 32) and 0.0 <= float(out.min()) and float(out.max()) <= 1.0 and tg == {"t": 1}
    batch = torch.rand(8, 3, 32, 32, device="cuda")
    out, _, _ = pipe((batch, None, None))
    assert out.shape == batch.shape and not torch.equal(out, batch) and bool(torch.isfinite(out).all())
    path = str(tmp_path / "syn.flat")
    write_synthetic_flat(path, 32, 32, {"taxa_L10": 5}, seed=3)
    ds = FlatSyntheticDataset(path)
    raw_batches = list(DevicePrefetcher(FlatBatchLoader(ds, 16, raw_uint8=True)))
    f32_batches = list(FlatBatchLoader(ds, 16))
    for (raw, *_), (f32, *_) in zip(raw_batches, f32_batches):
        assert raw.is_cuda and raw.dtype == torch.uint8
        assert torch.equal(u8hwc_to_f32chw(raw).cpu(), f32)


@gpu
def test_erase_rects_ignores_rectangles_outside_the_batch_or_the_image():
    """lnx_erase_rects takes its rectangle list from device memory, so the host entry cannot validate it: a rectangle of an image
    index outside [0, B), an empty one, or pixels left / above / right / below the image must be skipped, not written."""
    import ctypes as C

    from linnaeus_amd import _lib as L

    B, Cn, H, W = 2, 3, 8, 8
    guard = torch.full((4, Cn, H, W), 0.5, device="cuda")  # images 1..2 are the batch, 0 and 3 the canaries around it
    x = guard[1:3]
    rects = torch.tensor([[5, 0, 0, 4, 4], [-1, 0, 0, 4, 4], [0, -2, -3, 4, 5], [1, 6, 6, 5, 5], [1, 2, 2, 0, 3]], dtype=torch.int32, device="cuda")
    vals = torch.arange(1, 1 + rects.shape[0] * Cn, dtype=torch.float32, device="cuda").reshape(-1, Cn)
    L.check(L.lib().lnx_erase_rects(C.c_void_p(x.data_ptr()), B, Cn, H, W, C.c_void_p(rects.data_ptr()), C.c_void_p(vals.data_ptr()), rects.shape[0],
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_erase_rects")
    torch.cuda.synchronize()
    want = torch.full((B, Cn, H, W), 0.5, device="cuda")
    want[0, :, 0:2, 0:2] = vals[2][:, None, None]  # rows -2..1, columns -3..1 clipped to the image
    want[1, :, 6:8, 6:8] = vals[3][:, None, None]
    assert torch.equal(x, want)
    assert torch.equal(guard[0], torch.full_like(guard[0], 0.5)) and torch.equal(guard[3], torch.full_like(guard[3], 0.5))
