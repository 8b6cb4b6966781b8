#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference itself*.

Runs ONLY in the build container, where /root/reference is mounted:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen/make_golden.py

It imports `linnaeus` from /root/reference (read-only) with two tiny stand-ins for absent
third-party modules (yacs.config.CfgNode, termcolor.colored -- tests/golden/gen/_stubs/,
our own code), builds mFormerV1 through the reference's own build_model(), loads the
deterministic name-keyed weights of oracle.mformer_oracle.seeded_fill, and records inputs
and reference outputs as small .npz files.  Nothing from the reference is copied: the
fixtures hold numbers only (inputs, outputs, checksums, gradient norms).

Cases:
  tiny_a   dims 32..256, depths (1,1)/(1,1), E=3 (TEMPORAL+SPATIAL), 2 Linear heads, img 64
  tiny_b   depths (2,1)/(2,1), E=4 (+ELEVATION), ONLY_LAST_CLS, img 96
  tiny_c   metadata inactive (E=1), ConditionalClassifier heads on a real TaxonomyTree (F3)
  tiny_dp  tiny_a in train mode with DROP_PATH_RATE=0.5 and recorded per-call masks
  tiny_drop tiny_a in train mode with DROP_RATE=0.2 / ATTN_DROP_RATE=0.1 and the recorded keep mask of every nn.Dropout call
  sm       the real mFormerV1_sm config at 224, B=2, 4 Linear heads (1000/300/80/20)
plus per-op known answers (cos table, LN variants, dwconv, softmax-attention, aggregate) and
  soft_ce     the reference's TaxonomyAwareLabelSmoothingCE (per-sample losses + logits gradient)
  train_step  two optimizer steps of tiny_a (CE loss, clip_grad_norm_, AdamW): losses, grad norms, deltas.
  hier_loss   weighted_hierarchical_loss (null masking, class weights, task weights) + build_taxonomy_smoothing_matrix.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", REPO]
sys.dont_write_bytecode = True

import logging  # noqa: E402
import warnings  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

warnings.filterwarnings("ignore")
logging.disable(logging.CRITICAL)

from linnaeus.config import get_default_config  # noqa: E402
from linnaeus.models import build_model  # noqa: E402
from linnaeus.utils.config_utils import load_config, merge_configs  # noqa: E402
from yacs.config import CfgNode as CN  # noqa: E402

from oracle import mformer_oracle as O  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED = 20251003


def base_cfg(img):
    cfg = get_default_config()
    arch = load_config("/root/reference/configs/model/archs/mFormerV1/mFormerV1_sm.yaml")
    cfg.MODEL = merge_configs(cfg.MODEL, arch.MODEL)
    cfg.MODEL.IMG_SIZE = img
    cfg.MODEL.USE_FLASH_ATTN = False
    cfg.TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS = False
    cfg.MODEL.DROP_PATH_RATE = 0.0
    return cfg


def apply_spec(cfg, spec: O.Spec, head_type="Linear"):
    cfg.MODEL.CONVNEXT_STAGES.DIMS = list(spec.conv_dims)
    cfg.MODEL.CONVNEXT_STAGES.DEPTHS = [spec.conv_depths[0], spec.conv_depths[1], 9, 3]
    cfg.MODEL.ROPE_STAGES.DIMS = list(spec.rope_dims)
    cfg.MODEL.ROPE_STAGES.DEPTHS = list(spec.rope_depths)
    cfg.MODEL.ROPE_STAGES.NUM_HEADS = list(spec.rope_heads)
    cfg.MODEL.ROPE_STAGES.MLP_RATIO = list(spec.mlp_ratio)
    cfg.MODEL.ONLY_LAST_CLS = spec.only_last_cls
    cfg.MODEL.DROP_PATH_RATE = spec.drop_path_rate
    names = [n for n, _ in spec.meta]
    cfg.DATA.META.ACTIVE = bool(names)
    for comp in ("TEMPORAL", "SPATIAL", "ELEVATION"):
        cfg.DATA.META.COMPONENTS[comp].ENABLED = comp in names
    for n, d in spec.meta:
        assert cfg.DATA.META.COMPONENTS[n].DIM == d
    tasks = [t for t, _ in spec.heads]
    cfg.DATA.TASK_KEYS_H5 = tasks
    if head_type == "Linear":
        cfg.MODEL.CLASSIFICATION.HEADS = CN({t: {"TYPE": "Linear"} for t in tasks})
    else:
        cfg.MODEL.CLASSIFICATION.HEADS = CN(
            {t: {"TYPE": head_type, "ROUTING_STRATEGY": "soft", "TEMPERATURE": 1.0, "USE_BIAS": True} for t in tasks}
        )
    return cfg


def checksum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().max().item(), t.mean().item(), (t * t).sum().sqrt().item()])


def first_slice(t, n=16):
    return t.detach().reshape(-1)[:n].float().numpy().copy()


def load_seeded(model, seed):
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "hmatrix" not in k:
            new[k] = O.seeded_fill(canonical_name(k), v.shape, seed)
        else:
            new[k] = v
    model.load_state_dict(new, strict=True)
    return {k: v.clone() for k, v in model.state_dict().items()}


def canonical_name(k):
    """Hierarchical heads alias the shared Linear under every task
    (head.{t}.level_classifiers.{t'}.*); key the fill by the target task only so all aliases
    agree, and so the same numbers land in head.{t'}.fc.* of the Linear-head layout."""
    parts = k.split(".")
    if parts[0] == "head" and len(parts) >= 5 and parts[2] == "level_classifiers":
        return f"head.{parts[3]}.fc.{parts[4]}"
    return k


TAP_MODULES = {
    "stem": "stem",
    "stage0": "stages.0.{last0}",
    "down0": "downsample_layers.0",
    "stage1": "stages.1.{last1}",
    "down1": "downsample_layers.1",
    "rope0": "stages.2.{last2}",
    "down2": "downsample_layers.2",
    "rope1": "stages.3.{last3}",
}


def run_case(name, spec, img, batch, head_type="Linear", taxonomy=None, train_drop=False, grads=True):
    cfg = apply_spec(base_cfg(img), spec, head_type)
    kwargs = {}
    if spec.heads:
        kwargs["num_classes"] = {t: c for t, c in spec.heads}
    if taxonomy is not None:
        kwargs["taxonomy_tree"] = taxonomy
    model = build_model(cfg, **kwargs)
    ref_sd = load_seeded(model, SEED)

    # the oracle's inventory must reproduce the reference's names/shapes/order (Linear layout)
    if head_type == "Linear":
        shapes = O.param_shapes(spec)
        assert list(shapes.keys()) == list(ref_sd.keys()), (
            [k for k in shapes if k not in ref_sd], [k for k in ref_sd if k not in shapes],
            [(a, b) for a, b in zip(shapes, ref_sd) if a != b][:5])
        for k in shapes:
            assert tuple(ref_sd[k].shape) == shapes[k], (k, ref_sd[k].shape, shapes[k])

    x, meta = O.seeded_inputs(spec, batch, img, SEED + 1)
    rec = {"x": x.numpy(), "img": np.array(img), "batch": np.array(batch)}
    if meta is not None:
        rec["meta"] = meta.numpy()

    taps = {}
    hooks = []
    mods = dict(model.named_modules())
    last = {f"last{i}": len(model.stages[i]) - 1 for i in range(4)}
    for tname, path in TAP_MODULES.items():
        m = mods[path.format(**last)]
        hooks.append(m.register_forward_hook(lambda mod, inp, out, tname=tname: taps.__setitem__(tname, out.detach())))

    drop_scales = None
    if train_drop:
        model.train()
        torch.manual_seed(777)
        # replay the draws DropPath makes, call by call (blocks/drop_path.py:31-32): one
        # rand(B,1,..) per call whose prob > 0 (prob==0 -> nn.Identity, no draw)
        probs = O.drop_call_probs(spec)
        drop_scales = []
        for p in probs:
            if p == 0.0:
                drop_scales.append(None)
            else:
                keep = 1.0 - p
                r = torch.rand(batch)
                drop_scales.append(torch.floor(keep + r) / keep)
        torch.manual_seed(777)
        for i, s in enumerate(drop_scales):
            if s is not None:
                rec[f"drop_scale_{i}"] = s.numpy()
    else:
        model.eval()

    for p_ in model.parameters():
        p_.requires_grad_(True)
    feats = model.forward_features(x, meta)
    for h in hooks:
        h.remove()
    rec["feats"] = feats.detach().numpy()
    for tname, v in taps.items():
        rec["tap_" + tname] = checksum(v)
        rec["tapslice_" + tname] = first_slice(v)

    if spec.heads:
        if train_drop:
            torch.manual_seed(777)
        out = model(x, meta)
        for t, lg in out.items():
            rec["logits_" + t] = lg.detach().numpy()
        if grads:
            loss = O.probe_loss(out)
            rec["loss"] = np.array(loss.item())
            model.zero_grad()
            loss.backward()
            seen = {}
            for k, p_ in model.named_parameters():  # remove_duplicate -> shared heads once
                if p_.grad is None:
                    continue
                ck = canonical_name(k)
                seen[ck] = p_.grad
            names = sorted(seen)
            rec["grad_names"] = np.array(names)
            rec["grad_norms"] = np.array([seen[k].double().norm().item() for k in names])
            rec["grad_sums"] = np.array([seen[k].double().sum().item() for k in names])
            for k in names:
                rec["gradslice_" + k] = first_slice(seen[k], 8)

    # cross-check the oracle right here, against the live reference (pins the restatement)
    osd = {canonical_name(k): v for k, v in ref_sd.items() if "hmatrix" not in k}
    with torch.no_grad():
        ofe = O.forward_features(osd, spec, x, meta, drop_scales)
    err = (ofe - feats.detach()).abs().max().item()
    print(f"[{name}] params={sum(p.numel() for p in model.parameters()):,} oracle-vs-reference max|dfeats| = {err:.3e}")
    assert err < 5e-5, err
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)
    return model


def run_dropout_case(name, spec, img, batch, drop_rate=0.2, attn_drop_rate=0.1):
    """MODEL.DROP_RATE / MODEL.ATTN_DROP_RATE > 0 in training mode on the reference model (blocks/mlp.py:61-66,
    rope_2d_mhsa.py:497,503).  Every nn.Dropout of the reference goes through torch.nn.functional.dropout: for the duration
    of the run that function is replaced by one that draws its keep mask from a seeded generator, records it and applies
    x * mask / keep (the definition of dropout), so the fixture holds the masks the reference used, call by call -- per RoPE
    block: attention probabilities [B, h, N, N], proj_drop [B, N, C], Mlp hidden [B, N, hidden], Mlp output [B, N, C] --
    next to its logits, loss and gradients."""
    import torch.nn.functional as F

    cfg = apply_spec(base_cfg(img), spec, "Linear")
    cfg.MODEL.DROP_RATE = drop_rate
    cfg.MODEL.ATTN_DROP_RATE = attn_drop_rate
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    ref_sd = load_seeded(model, SEED)
    model.train()
    x, meta = O.seeded_inputs(spec, batch, img, SEED + 1)
    rec = {"x": x.numpy(), "meta": meta.numpy(), "img": np.array(img), "batch": np.array(batch),
           "drop_rate": np.array(drop_rate), "attn_drop_rate": np.array(attn_drop_rate)}
    masks = []
    gen = torch.Generator().manual_seed(SEED + 21)
    real = F.dropout

    def recorded_dropout(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        keep = 1.0 - p
        m = torch.rand(input.shape, generator=gen) < keep
        masks.append((m, float(p)))
        return input * m.to(input.dtype) / keep

    F.dropout = recorded_dropout
    try:
        out = model(x, meta)
        loss = O.probe_loss(out)
        model.zero_grad()
        loss.backward()
    finally:
        F.dropout = real
    n_blocks = sum(spec.rope_depths)
    assert len(masks) == 4 * n_blocks, (len(masks), n_blocks)
    for i, (m, p_) in enumerate(masks):
        rec[f"mask_{i}"] = np.packbits(m.numpy().reshape(-1))
        rec[f"mask_shape_{i}"] = np.array(m.shape)
        rec[f"mask_p_{i}"] = np.array(p_)
    rec["n_masks"] = np.array(len(masks))
    for t, lg in out.items():
        rec["logits_" + t] = lg.detach().numpy()
    rec["loss"] = np.array(loss.item())
    seen = {canonical_name(k): p_.grad for k, p_ in model.named_parameters() if p_.grad is not None}
    names = sorted(seen)
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array([seen[k].double().norm().item() for k in names])
    rec["grad_sums"] = np.array([seen[k].double().sum().item() for k in names])
    for k in names:
        rec["gradslice_" + k] = first_slice(seen[k], 8)
    # pin the oracle's dropout restatement against the live reference, with the recorded masks
    osd = {canonical_name(k): v.clone().requires_grad_(True) for k, v in ref_sd.items() if "hmatrix" not in k}
    mult = O.dropout_multipliers_from_masks([m for m, _ in masks], [p_ for _, p_ in masks])
    oout = O.forward(osd, spec, x, meta, None, dropout=mult)
    err = max((oout[t].detach() - out[t].detach()).abs().max().item() for t in out)
    O.probe_loss(oout).backward()
    gnum = sum((osd[k].grad - seen[k]).double().pow(2).sum().item() for k in names)
    gden = sum(seen[k].double().pow(2).sum().item() for k in names)
    gerr = (gnum / gden) ** 0.5  # global relative gradient error (a per-tensor ratio blows up on tensors whose gradient is ~0)
    print(f"[{name}] {len(masks)} dropout calls recorded; oracle-vs-reference max|dlogits| = {err:.3e}, global relative gradient error {gerr:.3e}")
    assert err < 5e-5 and gerr < 1e-4, (err, gerr)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def run_train_step(name, spec, img, batch, steps=2):
    """Caller (ii) of SURVEY 8c: forward -> per-task mean CE summed with static weights -> backward ->
    clip_grad_norm_ -> AdamW step (the sequence of train.py:147-176,279-316 without AMP scaling), on
    the reference model.  Records loss, pre-clip total gradient norm and parameter deltas per step."""
    cfg = apply_spec(base_cfg(img), spec, "Linear")
    model = build_model(cfg, num_classes={t: c for t, c in spec.heads})
    load_seeded(model, SEED)
    model.train()  # DROP_PATH_RATE 0 -> deterministic
    x, meta = O.seeded_inputs(spec, batch, img, SEED + 1)
    g = torch.Generator().manual_seed(SEED + 2)
    targets = {t: torch.randint(0, c, (batch,), generator=g) for t, c in spec.heads}
    weights = {t: 1.0 / (i + 1) for i, (t, _) in enumerate(spec.heads)}
    rec = {"x": x.numpy(), "meta": meta.numpy(), "steps": np.array(steps), "lr": np.array(1e-3), "wd": np.array(0.05),
           "clip": np.array(1.0), "task_weights": np.array([weights[t] for t, _ in spec.heads])}
    for t, _ in spec.heads:
        rec["target_" + t] = targets[t].numpy()
    params = {canonical_name(k): p_ for k, p_ in model.named_parameters()}
    before = {k: v.detach().clone() for k, v in params.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8)
    for s in range(steps):
        out = model(x, meta)
        loss = sum(weights[t] * torch.nn.functional.cross_entropy(out[t].float(), targets[t]) for t, _ in spec.heads)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        total = torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        opt.step()
        rec[f"loss_{s}"] = np.array(loss.item())
        rec[f"gnorm_{s}"] = np.array(float(total))
    names = sorted(params)
    rec["param_names"] = np.array(names)
    rec["delta_norms"] = np.array([(params[k].detach() - before[k]).double().norm().item() for k in names])
    for k in names:
        rec["deltaslice_" + k] = first_slice(params[k].detach() - before[k], 8)
    print(f"[{name}] losses {[float(rec[f'loss_{s}']) for s in range(steps)]} gnorms {[float(rec[f'gnorm_{s}']) for s in range(steps)]}")
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **rec)


def soft_ce_known_answers():
    """TaxonomyAwareLabelSmoothingCE of the reference (loss/taxonomy_label_smoothing.py) on seeded inputs: per-sample
    losses and the gradient of a weighted sum of them, with and without ignore_index / class weights."""
    from linnaeus.loss.taxonomy_label_smoothing import TaxonomyAwareLabelSmoothingCE

    g = torch.Generator().manual_seed(SEED + 7)
    B, Cn = 24, 37
    logits = (torch.randn(B, Cn, generator=g) * 3).requires_grad_(True)
    target = torch.randint(0, Cn, (B,), generator=g)
    target[:5] = 0  # null class rows
    soft = torch.rand(Cn, Cn, generator=g) * 0.1
    soft[torch.arange(Cn), torch.arange(Cn)] += 5.0
    soft = soft / soft.sum(1, keepdim=True)
    cw = torch.rand(Cn, generator=g) + 0.5
    wsum = torch.rand(B, generator=g)
    rec = {"logits": logits.detach().numpy(), "target": target.numpy(), "soft": soft.numpy(), "class_weight": cw.numpy(), "wsum": wsum.numpy()}
    for name, kw in (("plain", {}), ("ignore0", {"ignore_index": 0}), ("ignore0_cw", {"ignore_index": 0, "weight": cw, "apply_class_weights": True}),
                     ("cw", {"weight": cw, "apply_class_weights": True})):
        crit = TaxonomyAwareLabelSmoothingCE(soft, **kw)
        logits.grad = None
        loss = crit(logits, target)
        (loss * wsum).sum().backward()
        rec[f"loss_{name}"] = loss.detach().numpy()
        rec[f"grad_{name}"] = logits.grad.numpy().copy()
        print(f"[soft_ce/{name}] mean loss {loss.mean().item():.5f}")
    np.savez_compressed(os.path.join(OUT, "soft_ce.npz"), **rec)


def hier_loss_known_answers():
    """The reference's whole loss path on seeded logits (loss/hierarchical_loss.py:24-406 with loss/masking.py,
    loss/gradient_weighting.py:301-358 and loss/taxonomy_label_smoothing.py:30-130): total loss, per-task weighted
    losses and the gradient wrt every task's logits, for the deterministic masking modes (scheduled with inclusion
    probability 1 and 0, PHASE1_MASK_NULL_LOSS, validation), with and without class weights (finding F13: the weights
    enter three times on the scheduled path, twice on the PHASE1 path).  Also build_taxonomy_smoothing_matrix known
    answers (root rows, disconnected classes, beta)."""
    from linnaeus.loss.gradient_weighting import GradientWeighting
    from linnaeus.loss.hierarchical_loss import weighted_hierarchical_loss
    from linnaeus.loss.taxonomy_label_smoothing import TaxonomyAwareLabelSmoothingCE, build_taxonomy_smoothing_matrix

    g = torch.Generator().manual_seed(SEED + 11)
    tasks = (("taxa_L10", 23), ("taxa_L20", 9), ("taxa_L30", 4))
    B = 16
    rec = {"tasks": np.array([t for t, _ in tasks]), "classes": np.array([c for _, c in tasks])}
    # smoothing matrices from synthetic tree distances (every rank: a distance matrix with some inf entries)
    soft = {}
    for t, c in tasks:
        d = torch.randint(1, 5, (c, c), generator=g).float() * 2.0
        d = torch.minimum(d, d.t())
        d.fill_diagonal_(0.0)
        if c > 5:
            d[2, 5] = d[5, 2] = float("inf")
            d[c - 1, :] = float("inf")  # a class disconnected from everything: uniform fallback row
            d[:, c - 1] = float("inf")
            d[c - 1, c - 1] = 0.0
        roots = [0, 3] if c > 3 else [0]
        for beta, ur in ((1.0, True), (0.5, False)):
            m = build_taxonomy_smoothing_matrix(c, d, alpha=0.15, beta=beta, uniform_roots=ur, root_class_ids=roots)
            rec[f"smooth_{t}_b{beta}_u{int(ur)}"] = m.numpy()
        rec[f"dist_{t}"] = d.numpy()
        rec[f"roots_{t}"] = np.array(roots)
        soft[t] = build_taxonomy_smoothing_matrix(c, d, alpha=0.15, beta=1.0, uniform_roots=True, root_class_ids=roots)
    logits = {t: (torch.randn(B, c, generator=g) * 2).requires_grad_(True) for t, c in tasks}
    targets = {t: torch.randint(0, c, (B,), generator=g) for t, c in tasks}
    targets["taxa_L10"][:3] = 0
    targets["taxa_L20"][2:7] = 0
    cw = {t: {i: float(0.5 + torch.rand(1, generator=g).item()) for i in range(0, c, 2)} for t, c in tasks}  # sparse dict: missing -> 1.0
    tw = {"taxa_L10": 1.0, "taxa_L20": 0.6, "taxa_L30": 0.3}
    for t, c in tasks:
        rec[f"logits_{t}"] = logits[t].detach().numpy()
        rec[f"target_{t}"] = targets[t].numpy()
        rec[f"soft_{t}"] = soft[t].numpy()
        v = np.ones(c, dtype=np.float32)
        for i, w in cw[t].items():
            v[i] = w
        rec[f"cw_{t}"] = v
    rec["task_weights"] = np.array([tw[t] for t, _ in tasks], dtype=np.float32)

    class Sched:  # the one method the loss path calls (ops_schedule/ops_schedule.py:655)
        def __init__(self, p):
            self.p = p

        def get_null_mask_prob(self, step):
            return self.p

    criteria = {t: TaxonomyAwareLabelSmoothingCE(soft[t]) for t, _ in tasks}
    keys = [t for t, _ in tasks]
    for mode, prob, phase1, val, use_cw in (("sched1", 1.0, False, False, True), ("sched0", 0.0, False, False, True), ("phase1", 1.0, True, False, True),
                                            ("val", 0.0, False, True, True), ("sched0_nocw", 0.0, False, False, False)):
        cfg = get_default_config()
        cfg.defrost()
        cfg.TRAIN.PHASE1_MASK_NULL_LOSS = phase1
        gw = GradientWeighting(keys, cfg, "static", init_weights=tw, class_weights=cw if use_cw else None)
        for t in keys:
            logits[t].grad = None
        total, comps, weights = weighted_hierarchical_loss({t: logits[t] for t in keys}, targets, criteria, gw, Sched(prob), 10,
                                                           is_validation=val, config=cfg)
        total.backward()
        rec[f"{mode}_total"] = np.float64(total.item())
        rec[f"{mode}_weighted"] = np.array([comps["weighted_tasks"][t] for t in keys])
        rec[f"{mode}_masked_mean"] = np.array([comps["masked_tasks"][t] for t in keys])
        rec[f"{mode}_raw_mean"] = np.array([comps["tasks"][t] for t in keys])
        for t in keys:
            rec[f"{mode}_grad_{t}"] = logits[t].grad.numpy().copy()
        print(f"[hier_loss/{mode}] total {total.item():.6f} weighted {rec[f'{mode}_weighted']}")
    np.savez_compressed(os.path.join(OUT, "hier_loss.npz"), **rec)


def muon_known_answers():
    """The reference's Muon (optimizers/muon.py): Newton-Schulz orthogonalisation of seeded matrices (wide, tall, a
    conv-shaped one) and two optimizer steps on three parameters."""
    from linnaeus.optimizers.muon import Muon, zeropower_via_newtonschulz5

    g = torch.Generator().manual_seed(SEED + 13)
    rec = {}
    for name, shape in (("wide", (48, 192)), ("tall", (192, 48)), ("odd", (40, 100)), ("sq", (64, 64))):
        G = torch.randn(*shape, generator=g)
        rec[f"ns_in_{name}"] = G.numpy()
        rec[f"ns_out_{name}"] = zeropower_via_newtonschulz5(G, steps=5).float().numpy()
    ps = [torch.randn(48, 192, generator=g), torch.randn(192, 48, generator=g), torch.randn(32, 8, 2, 2, generator=g)]
    params = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = Muon(params, lr=0.02, weight_decay=0.01, momentum=0.95, nesterov=True, ns_steps=5)
    for i, p in enumerate(ps):
        rec[f"p{i}_init"] = p.numpy()
    for step in range(2):
        for i, p in enumerate(params):
            gr = torch.randn(p.shape, generator=g)
            rec[f"p{i}_grad{step}"] = gr.numpy()
            p.grad = gr.clone()
        opt.step()
    for i, p in enumerate(params):
        rec[f"p{i}_final"] = p.detach().numpy()
    print("[muon] orthogonality |X X^T - I| (wide):", float((torch.from_numpy(rec["ns_out_wide"]) @ torch.from_numpy(rec["ns_out_wide"]).t() - torch.eye(48)).abs().max()))
    np.savez_compressed(os.path.join(OUT, "muon.npz"), **rec)


def collate_known_answers():
    """The reference's GPU augmentations on a seeded batch (they are device-agnostic torch code): GPUSelectiveMixup and
    GPUSelectiveCutMix, with the random draws they made (permutation, lambda, box, per-sample pick) recorded so the HIP
    path can be replayed with the same draws."""
    import linnaeus.aug.gpu.selective_cutmix as cm
    from linnaeus.aug.gpu.selective_cutmix import GPUSelectiveCutMix
    from linnaeus.aug.gpu.selective_mixup import GPUSelectiveMixup

    g = torch.Generator().manual_seed(SEED + 17)
    B, Cc, H, W, D = 12, 3, 16, 24, 15
    bounds = [(0, 2), (2, 5), (5, 15)]
    images = torch.rand(B, Cc, H, W, generator=g)
    tasks = (("taxa_L10", 9), ("taxa_L20", 4))
    lab = {t: torch.randint(0, c, (B,), generator=g) for t, c in tasks}
    lab["taxa_L10"][1] = 0  # a null sample: excluded from mixing
    targets = {t: torch.nn.functional.one_hot(lab[t], c).float() for t, c in tasks}
    aux = torch.randn(B, D, generator=g)
    aux[2, 0:2] = 0.0       # absent chunk
    aux[3, 3] = 0.0         # partially zero chunk -> treated as absent
    aux[5, 5:15] = 0.0
    aux[7] = 0.0
    masks = aux != 0.0
    gids = torch.tensor([0, 0, 0, 1, 1, 1, 1, 2, 3, 3, -1, 0])
    rec = {"images": images.numpy(), "aux": aux.numpy(), "masks": masks.numpy(), "gids": gids.numpy(), "bounds": np.array(bounds)}
    for t, _ in tasks:
        rec[f"target_{t}"] = targets[t].numpy()
    draws = {}
    orig_rand, orig_beta = torch.rand, torch.distributions.beta.Beta.sample

    def rand_spy(*a, **k):
        r = orig_rand(*a, **k)
        if r.numel() == B:
            draws["pick"] = r.clone()
        return r

    def beta_spy(self, *a, **k):
        r = orig_beta(self, *a, **k)
        draws["lam"] = float(r)
        return r

    orig_bbox = cm.rand_bbox

    def bbox_spy(size, lam):
        r = orig_bbox(size, lam)
        draws["box"] = r
        return r

    torch.rand, torch.distributions.beta.Beta.sample, cm.rand_bbox = rand_spy, beta_spy, bbox_spy
    try:
        for name, cls, cfgd in (("mixup", GPUSelectiveMixup, {"PROB": 1.0, "ALPHA": 0.4, "meta_chunk_bounds_list": bounds}),
                                ("cutmix", GPUSelectiveCutMix, {"PROB": 1.0, "ALPHA": 1.0, "meta_chunk_bounds_list": bounds})):
            torch.manual_seed(SEED + (3 if name == "mixup" else 5))
            import random as _r
            _r.seed(SEED)
            op = cls(cfgd, config=None)
            perm_holder = {}
            orig_perm = op._get_ingroup_permutation

            def perm_spy(gi, _o=orig_perm, _h=perm_holder):
                p_ = _o(gi)
                _h["perm"] = p_.clone()
                return p_

            op._get_ingroup_permutation = perm_spy
            out = op((images.clone(), {k: v.clone() for k, v in targets.items()}, aux.clone(), masks.clone(), gids.clone()), exclude_null_samples=True)
            rec[f"{name}_perm"] = perm_holder["perm"].numpy()
            rec[f"{name}_lam"] = np.float64(draws["lam"])
            rec[f"{name}_pick"] = draws["pick"].numpy()
            if name == "cutmix":
                rec["cutmix_box"] = np.array(draws["box"])
            rec[f"{name}_images"] = out[0].numpy()
            for t, _ in tasks:
                rec[f"{name}_target_{t}"] = out[1][t].numpy()
            rec[f"{name}_aux"] = out[2].numpy()
            rec[f"{name}_masks"] = out[3].numpy()
            print(f"[collate/{name}] lam {draws['lam']:.4f} perm {perm_holder['perm'].tolist()} box {draws.get('box')}")
    finally:
        torch.rand, torch.distributions.beta.Beta.sample, cm.rand_bbox = orig_rand, orig_beta, orig_bbox
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **rec)


def aug_known_answers():
    """SURVEY 8f-3 remainder: outputs of the reference's GPU augmentation classes on CPU tensors, for what of them runs.
    GPUAutoAugmentBatch._apply_op for Posterize / PosterizeOriginal / PosterizeIncreasing / Solarize / SolarizeAdd / Invert
    (autoaug.py:86,117-138), the whole __call__ (:88-103, CPU coin flips) on a policy made of those operations, and
    GPURandomErasing (random_erasing.py:24-94) at B = 1 -- the batch size of its per-sample pipeline and the only one its
    `torch.randint(0, <tensor>, ...)` accepts -- in all three modes, with the draws it made re-derived from the same seed.  The
    other AutoAugment operations raise upstream (recorded in `raises`), so there is nothing to record for them."""
    from linnaeus.aug.gpu.autoaug import GPUAutoAugmentBatch
    from linnaeus.aug.gpu.random_erasing import GPURandomErasing
    from oracle import aug_oracle as A

    g = torch.Generator().manual_seed(SEED + 31)
    img = torch.rand(3, 3, 24, 20, generator=g)
    rec = {"img": img.numpy()}
    aa = GPUAutoAugmentBatch("v0r", 0.4)
    working, raising = [], []
    for op, mags in (("Posterize", (8, 5)), ("PosterizeOriginal", (7, 6)), ("PosterizeIncreasing", (8, 2, 6)), ("Solarize", (5, 3, 10)), ("SolarizeAdd", (3, 7)),
                     ("Invert", (4,))):
        for m in mags:
            out = aa._apply_op(img.clone(), op, m)
            rec[f"op_{op}_{m}"] = out.numpy()
            working.append(f"{op}:{m}")
    for op in ("ShearX", "ShearY", "TranslateX", "TranslateY", "TranslateYRel", "Rotate", "Color", "Contrast", "Sharpness", "Brightness", "AutoContrast", "Equalize",
               "Desaturate", "GaussianBlurRand"):
        try:
            aa._apply_op(img.clone(), op, 5)
        except Exception as e:  # AttributeError (torch.nn.functional has no affine / rotate / adjust_* / gaussian_blur) or TypeError (arity)
            raising.append(f"{op}:{type(e).__name__}")
    rec["working_ops"] = np.array(working)
    rec["raising_ops"] = np.array(raising)
    # pinned restatements
    for key in working:
        op, m = key.split(":")
        mm = int(m) * 0.1
        want = {"Posterize": lambda: A.posterize(img, mm), "PosterizeOriginal": lambda: A.posterize(img, mm), "PosterizeIncreasing": lambda: A.posterize(img, 8 - mm),
                "Solarize": lambda: A.solarize(img, mm), "SolarizeAdd": lambda: A.solarize_add(img, mm), "Invert": lambda: A.invert(img)}[op]()
        assert torch.equal(want, torch.from_numpy(rec[f"op_{op}_{m}"])), key
    aa.policy = [[("Solarize", 0.6, 5), ("Invert", 0.5, 4)], [("PosterizeIncreasing", 0.7, 6), ("SolarizeAdd", 0.8, 3)], [("Posterize", 0.4, 8), ("Solarize", 0.6, 3)],
                 [("Invert", 0.2, 1)], [("SolarizeAdd", 0.9, 7), ("PosterizeOriginal", 0.5, 6)]]
    rec["call_policy"] = np.array([" ".join(f"{o}:{p_}:{m}" for o, p_, m in sub) for sub in aa.policy])
    for seed in (3, 4, 5):
        torch.manual_seed(seed)
        rec[f"call_seed{seed}"] = aa((img * 1.3 - 0.1).clone()).numpy()  # input outside [0, 1]: the initial clamp matters
    # random erasing, B = 1
    img1 = torch.rand(1, 3, 40, 36, generator=g)
    rec["re_img"] = img1.numpy()
    for mode in ("const", "rand", "pixel"):
        cfg = {"PROB": 0.9, "AREA_RANGE": [0.02, 0.3], "ASPECT_RATIO": [0.3, 3.3], "COUNT": 2, "MODE": mode}
        re = GPURandomErasing(cfg)
        for seed in (11, 12):
            torch.manual_seed(seed)
            out = re(img1.clone())
            # the same draws again (one valid sample: rand(1); per COUNT uniform_, uniform_, randint, randint, uniform_ / randn)
            torch.manual_seed(seed)
            H, W = img1.shape[2:]
            draws = [{"gate": torch.rand(1)}]
            if float(draws[0]["gate"]) <= cfg["PROB"]:
                for it in range(cfg["COUNT"]):
                    d = draws[it] if it == 0 else {}
                    d["areas"] = torch.empty(1).uniform_(cfg["AREA_RANGE"][0] * H * W, cfg["AREA_RANGE"][1] * H * W)
                    d["aspects"] = torch.empty(1).uniform_(*cfg["ASPECT_RATIO"])
                    h = torch.sqrt(d["areas"] * d["aspects"]).round().long()
                    w = torch.sqrt(d["areas"] / d["aspects"]).round().long()
                    if bool(((w < W) & (h < H)).any()):
                        d["x"] = torch.randint(0, int(W - w), (1,))
                        d["y"] = torch.randint(0, int(H - h), (1,))
                        d["values"] = torch.randn(1, 3, 1, 1) if mode == "pixel" else torch.empty(1, 3, 1, 1).uniform_(0, 1)
                    else:
                        d["x"] = d["y"] = torch.zeros(1, dtype=torch.long)
                        d["values"] = torch.zeros(1, 3, 1, 1)
                    if it > 0:
                        draws.append(d)
            else:
                for it in range(cfg["COUNT"]):
                    d = draws[it] if it == 0 else {}
                    d.update(areas=torch.ones(1), aspects=torch.ones(1), x=torch.zeros(1, dtype=torch.long), y=torch.zeros(1, dtype=torch.long), values=torch.zeros(1, 3, 1, 1))
                    if it > 0:
                        draws.append(d)
            again = A.random_erasing(img1, draws, cfg)
            assert torch.equal(again, out), (mode, seed, (again - out).abs().max())
            rec[f"re_{mode}_{seed}_out"] = out.numpy()
            rec[f"re_{mode}_{seed}_gate"] = draws[0]["gate"].numpy()
            for it, d in enumerate(draws):
                for k in ("areas", "aspects", "x", "y", "values"):
                    rec[f"re_{mode}_{seed}_{it}_{k}"] = d[k].numpy()
    rec["re_cfg"] = np.array([0.9, 0.02, 0.3, 0.3, 3.3, 2])
    print(f"[aug] reference ops that run: {working}; ops that raise upstream: {raising}")
    np.savez_compressed(os.path.join(OUT, "aug.npz"), **rec)


def per_op_known_answers():
    """Small known-answer vectors produced by the reference's own functions/modules."""
    from linnaeus.models.blocks.convnext import ConvNeXtBlock, ConvNeXtDownsampleLayer, LayerNormChannelsFirst
    from linnaeus.models.blocks.rope_2d_mhsa import RoPE2DAttention, RoPE2DMHSABlock, apply_rotary_emb, compute_mixed_cis, init_t_xy
    from linnaeus.models.normalization import ResNormLayer

    g = torch.Generator().manual_seed(SEED + 7)
    rec = {}
    # cos table (F1)
    freqs = O.seeded_fill("x.attn.freqs", (2, 2, 32), SEED)
    tx, ty = init_t_xy(5, 3)
    cis = compute_mixed_cis(freqs, tx, ty).to(torch.float32)
    rec["rope_freqs"] = freqs.numpy()
    rec["rope_cos_3x5"] = cis.numpy()
    q = torch.randn(2, 2, 15, 64, generator=g)
    k = torch.randn(2, 2, 15, 64, generator=g)
    qr, kr = apply_rotary_emb(q, k, cis)
    rec.update(rope_q=q.numpy(), rope_k=k.numpy(), rope_q_out=qr.numpy(), rope_k_out=kr.numpy())
    # LayerNormChannelsFirst
    ln = LayerNormChannelsFirst(8, eps=1e-6)
    ln.weight.data = O.seeded_fill("a.norm.weight", (8,), SEED)
    ln.bias.data = O.seeded_fill("a.norm.bias", (8,), SEED)
    xi = torch.randn(2, 8, 3, 5, generator=g)
    rec.update(lncf_x=xi.numpy(), lncf_w=ln.weight.data.numpy(), lncf_b=ln.bias.data.numpy(), lncf_y=ln(xi).detach().numpy())
    # ConvNeXtBlock (border taps included: 6x5 image is smaller than the 7x7 window)
    blk = ConvNeXtBlock(8, drop_path=0.0, layer_scale_init_value=1.0).eval()
    bsd = {kk: O.seeded_fill("stages.0.0." + kk, v.shape, SEED) for kk, v in blk.state_dict().items()}
    blk.load_state_dict(bsd)
    xi = torch.randn(2, 8, 6, 5, generator=g)
    rec.update(cnb_x=xi.numpy(), cnb_y=blk(xi).detach().numpy(), cnb_dw=blk.dwconv(xi).detach().numpy())
    # Downsample
    dsl = ConvNeXtDownsampleLayer(8, 16).eval()
    dsd = {kk: O.seeded_fill("downsample_layers.0." + kk, v.shape, SEED) for kk, v in dsl.state_dict().items()}
    dsl.load_state_dict(dsd)
    xi = torch.randn(2, 8, 6, 4, generator=g)
    rec.update(ds_x=xi.numpy(), ds_y=dsl(xi).detach().numpy())
    # RoPE attention + block, E=3, grid 3x5, dim 128 (2 heads of 64)
    att = RoPE2DAttention(128, (3, 5), extra_token_num=3, num_heads=2, qkv_bias=True).eval()
    asd = {kk: O.seeded_fill("stages.2.0.attn." + kk, v.shape, SEED) for kk, v in att.state_dict().items()}
    att.load_state_dict(asd)
    xi = torch.randn(2, 18, 128, generator=g)
    rec.update(att_x=xi.numpy(), att_y=att(xi, 3, 5).detach().numpy())
    rb = RoPE2DMHSABlock(128, (3, 5), num_heads=2, qkv_bias=True, extra_token_num=3).eval()
    rsd = {kk: O.seeded_fill("stages.2.0." + kk, v.shape, SEED) for kk, v in rb.state_dict().items()}
    rb.load_state_dict(rsd)
    rec.update(rb_y=rb(xi, 3, 5).detach().numpy())
    # ResNormLayer inside a meta head
    seq = torch.nn.Sequential(torch.nn.Linear(3, 32), torch.nn.ReLU(inplace=True), torch.nn.LayerNorm(32), ResNormLayer(32)).eval()
    msd = {kk: O.seeded_fill("meta_spatial_head_1." + kk, v.shape, SEED) for kk, v in seq.state_dict().items()}
    seq.load_state_dict(msd)
    mi = torch.randn(4, 3, generator=g)
    rec.update(mh_x=mi.numpy(), mh_y=seq(mi).detach().numpy())
    np.savez_compressed(os.path.join(OUT, "per_op.npz"), **rec)
    print("[per_op] written")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])  # optional: regenerate just the named fixtures

    def want(n):
        return not only or n in only

    tiny_dims = (32, 64, 128, 256)
    heads2 = (("taxa_L10", 7), ("taxa_L20", 5))
    tiny_a = O.Spec(conv_dims=tiny_dims, conv_depths=(1, 1), rope_depths=(1, 1), rope_heads=(2, 4), heads=heads2)
    if want("tiny_a"):
        run_case("tiny_a", tiny_a, 64, 2)
    if want("tiny_b"):
        tiny_b = O.Spec(conv_dims=tiny_dims, conv_depths=(2, 1), rope_depths=(2, 1), rope_heads=(2, 4),
                        meta=(("TEMPORAL", 2), ("SPATIAL", 3), ("ELEVATION", 10)), only_last_cls=True, heads=heads2)
        run_case("tiny_b", tiny_b, 96, 3)
    if want("tiny_c"):
        from linnaeus.utils.taxonomy.taxonomy_tree import TaxonomyTree
        h3 = (("taxa_L10", 6), ("taxa_L20", 3), ("taxa_L30", 2))
        tree = TaxonomyTree(
            {"taxa_L10": {0: 0, 1: 0, 2: 1, 3: 1, 4: 2, 5: 2}, "taxa_L20": {0: 0, 1: 0, 2: 1}},
            [t for t, _ in h3], {t: c for t, c in h3})
        tiny_c = O.Spec(conv_dims=tiny_dims, conv_depths=(1, 1), rope_depths=(1, 1), rope_heads=(2, 4), meta=(), heads=h3)
        run_case("tiny_c", tiny_c, 64, 2, head_type="ConditionalClassifier", taxonomy=tree)
    if want("tiny_dp"):
        tiny_dp = O.Spec(conv_dims=tiny_dims, conv_depths=(2, 1), rope_depths=(2, 1), rope_heads=(2, 4), heads=heads2, drop_path_rate=0.5)
        run_case("tiny_dp", tiny_dp, 64, 4, train_drop=True)
    if want("sm"):
        sm = O.Spec(heads=(("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20)))
        run_case("sm", sm, 224, 2)
    if want("tiny_drop"):
        run_dropout_case("tiny_drop", tiny_a, 64, 2)
    if want("per_op"):
        per_op_known_answers()
    if want("soft_ce"):
        soft_ce_known_answers()
    if want("train_step"):
        run_train_step("train_step", tiny_a, 64, 4)
    if want("hier_loss"):
        hier_loss_known_answers()
    if want("muon"):
        muon_known_answers()
    if want("collate"):
        collate_known_answers()
    if want("aug"):
        aug_known_answers()


if __name__ == "__main__":
    main()
