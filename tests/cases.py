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
