"""Architecture specs of the golden cases (must match tests/golden/gen/make_golden.py)."""
from oracle import mformer_oracle as O

SEED = 20251003
TINY_DIMS = (32, 64, 128, 256)
HEADS2 = (("taxa_L10", 7), ("taxa_L20", 5))
HEADS3 = (("taxa_L10", 6), ("taxa_L20", 3), ("taxa_L30", 2))
HEADS_SM = (("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20))

CASES = {
    "tiny_a": O.Spec(conv_dims=TINY_DIMS, conv_depths=(1, 1), rope_depths=(1, 1), rope_heads=(2, 4), heads=HEADS2),
    "tiny_b": O.Spec(conv_dims=TINY_DIMS, conv_depths=(2, 1), rope_depths=(2, 1), rope_heads=(2, 4),
                     meta=(("TEMPORAL", 2), ("SPATIAL", 3), ("ELEVATION", 10)), only_last_cls=True, heads=HEADS2),
    "tiny_c": O.Spec(conv_dims=TINY_DIMS, conv_depths=(1, 1), rope_depths=(1, 1), rope_heads=(2, 4), meta=(), heads=HEADS3),
    "tiny_dp": O.Spec(conv_dims=TINY_DIMS, conv_depths=(2, 1), rope_depths=(2, 1), rope_heads=(2, 4), heads=HEADS2,
                      drop_path_rate=0.5),
    "sm": O.Spec(heads=HEADS_SM),
}


def load_case(name, golden_dir):
    import numpy as np
    import torch

    spec = CASES[name]
    z = np.load(f"{golden_dir}/{name}.npz", allow_pickle=False)
    x = torch.from_numpy(z["x"])
    meta = torch.from_numpy(z["meta"]) if "meta" in z.files else None
    sd = O.seeded_state_dict(O.param_shapes(spec), SEED)
    drops = None
    if spec.drop_path_rate > 0:
        drops = []
        for i in range(O.n_drop_calls(spec)):
            k = f"drop_scale_{i}"
            drops.append(torch.from_numpy(z[k]) if k in z.files else None)
    return spec, z, sd, x, meta, drops


def load_dropout_case(golden_dir, name="tiny_drop"):
    """tiny_drop.npz: the reference model (tiny_a architecture) in training mode with DROP_RATE / ATTN_DROP_RATE > 0 and the
    keep mask of every nn.Dropout call it made (make_golden.run_dropout_case).  Returns the masks in call order."""
    import numpy as np
    import torch

    spec = CASES["tiny_a"]
    z = np.load(f"{golden_dir}/{name}.npz", allow_pickle=False)
    x, meta = torch.from_numpy(z["x"]), torch.from_numpy(z["meta"])
    sd = O.seeded_state_dict(O.param_shapes(spec), SEED)
    masks, ps = [], []
    for i in range(int(z["n_masks"])):
        shape = tuple(int(v) for v in z[f"mask_shape_{i}"])
        n = int(np.prod(shape))
        masks.append(torch.from_numpy(np.unpackbits(z[f"mask_{i}"])[:n].reshape(shape).astype(np.bool_)))
        ps.append(float(z[f"mask_p_{i}"]))
    return spec, z, sd, x, meta, masks, ps


def plan_dropout_buffers(masks):
    """The recorded masks (per RoPE block: attention [B, h, N, N], proj [B, N, C], hidden [B, N, hid], fc2 [B, N, C]) in the
    layout lnx_plan_set_dropout / lnx_plan_set_attn_dropout take: per block proj | hidden | fc2 bytes, and per block the
    attention mask with its key axis padded to a multiple of 64."""
    import torch

    flat, attn = [], []
    for i in range(0, len(masks), 4):
        a, pj, hd, f2 = masks[i:i + 4]
        flat += [pj.reshape(-1).to(torch.uint8), hd.reshape(-1).to(torch.uint8), f2.reshape(-1).to(torch.uint8)]
        B, h, N, _ = a.shape
        Np = (N + 63) // 64 * 64
        pad = torch.zeros(B, h, N, Np, dtype=torch.uint8)
        pad[..., :N] = a.to(torch.uint8)
        attn.append(pad.reshape(-1))
    return torch.cat(flat), torch.cat(attn)


def make_config(spec, img, head_type="Linear", drop_path_rate=None):
    """linnaeus_amd config for a Spec (mirrors apply_spec() of the golden generator)."""
    from linnaeus_amd import default_config

    cfg = default_config()
    cfg.MODEL.IMG_SIZE = img
    cfg.MODEL.CONVNEXT_STAGES.DIMS = list(spec.conv_dims)
    cfg.MODEL.CONVNEXT_STAGES.DEPTHS = [spec.conv_depths[0], spec.conv_depths[1], 9, 3]
    cfg.MODEL.ROPE_STAGES.DIMS = list(spec.rope_dims)
    cfg.MODEL.ROPE_STAGES.DEPTHS = list(spec.rope_depths)
    cfg.MODEL.ROPE_STAGES.NUM_HEADS = list(spec.rope_heads)
    cfg.MODEL.ROPE_STAGES.MLP_RATIO = list(spec.mlp_ratio)
    cfg.MODEL.ONLY_LAST_CLS = spec.only_last_cls
    cfg.MODEL.DROP_PATH_RATE = spec.drop_path_rate if drop_path_rate is None else drop_path_rate
    names = [n for n, _ in spec.meta]
    cfg.DATA.META.ACTIVE = bool(names)
    for comp in ("TEMPORAL", "SPATIAL", "ELEVATION"):
        cfg.DATA.META.COMPONENTS[comp].ENABLED = comp in names
    tasks = [t for t, _ in spec.heads]
    cfg.DATA.TASK_KEYS_H5 = tasks
    if head_type == "Linear":
        cfg.MODEL.CLASSIFICATION.HEADS = {t: {"TYPE": "Linear"} for t in tasks}
    else:
        cfg.MODEL.CLASSIFICATION.HEADS = {t: {"TYPE": head_type, "ROUTING_STRATEGY": "soft", "TEMPERATURE": 1.0, "USE_BIAS": True} for t in tasks}
    return cfg


class TinyTree:
    """Stand-in with the one method the heads use (TaxonomyTree.build_hierarchy_matrices,
    utils/taxonomy/taxonomy_tree.py:384-405): matrices keyed "{parent}_{child}"."""

    def __init__(self, hierarchy_map, task_keys, num_classes):
        self.hierarchy_map, self.task_keys, self.num_classes = hierarchy_map, task_keys, num_classes

    def build_hierarchy_matrices(self):
        import torch

        out = {}
        for i in range(len(self.task_keys) - 1):
            child, parent = self.task_keys[i], self.task_keys[i + 1]
            m = torch.zeros(self.num_classes[parent], self.num_classes[child])
            for c, p_ in self.hierarchy_map.get(child, {}).items():
                m[p_, c] = 1.0
            out[f"{parent}_{child}"] = m
        return out


def model_state_dict_from_oracle(model, sd):
    """Map the oracle's Linear-layout state dict onto whatever head layout the model has."""
    out = {}
    for k, v in model.state_dict().items():
        parts = k.split(".")
        if parts[0] == "head" and len(parts) >= 5 and parts[2] == "level_classifiers":
            out[k] = sd[f"head.{parts[3]}.fc.{parts[4]}"]
        elif "hmatrix" in k:
            out[k] = v
        else:
            out[k] = sd[k]
    return out


def load_train_step(golden_dir):
    """Fixture of caller (ii), SURVEY 8c: two CE + clip + AdamW steps of tiny_a on the reference."""
    import numpy as np
    import torch

    spec = CASES["tiny_a"]
    z = np.load(f"{golden_dir}/train_step.npz", allow_pickle=False)
    sd = O.seeded_state_dict(O.param_shapes(spec), SEED)
    targets = {t: torch.from_numpy(z["target_" + t]) for t, _ in spec.heads}
    weights = {t: float(w) for (t, _), w in zip(spec.heads, z["task_weights"])}
    return spec, z, sd, torch.from_numpy(z["x"]), torch.from_numpy(z["meta"]), targets, weights


def train_steps(forward, params, z, spec, targets, weights):
    """The step sequence of train.py:147-176,279-316 (no AMP scaler): forward -> weighted per-task mean CE
    -> backward -> clip_grad_norm_ -> AdamW.  `forward()` returns {task: logits}; `params` is a list."""
    import torch

    opt = torch.optim.AdamW(params, lr=float(z["lr"]), weight_decay=float(z["wd"]), betas=(0.9, 0.999), eps=1e-8)
    losses, norms = [], []
    for _ in range(int(z["steps"])):
        out = forward()
        loss = sum(weights[t] * torch.nn.functional.cross_entropy(out[t].float(), targets[t].to(out[t].device)) for t, _ in spec.heads)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, float(z["clip"]))))
        opt.step()
        losses.append(loss.item())
    return losses, norms
