#!/usr/bin/env python3
"""Build-container only: time the IMPORTED reference mFormerV1_sm (fp32, CPU, all cores) next to the repo's CPU oracle on the
same inputs -- the cross-check SURVEY 8d asks for beside bench.py's `cpu_baseline` (which times the oracle on the GPU box,
where the reference does not exist).  Protocol: batch 8, 3 warm-up + 10 timed steps, forward-only (eval, no_grad) and
forward + 4-task CE + backward (train mode, DropPath 0, no checkpointing).  Writes profiles/r03_reference_cpu_timing.json.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/time_reference_cpu.py
"""
import json
import logging
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests", "golden", "gen", "_stubs"), "/root/reference", REPO]
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
logging.disable(logging.CRITICAL)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from yacs.config import CfgNode as CN  # noqa: E402

from linnaeus.config import get_default_config  # noqa: E402
from linnaeus.models import build_model  # noqa: E402
from linnaeus.utils.config_utils import load_config, merge_configs  # noqa: E402
from oracle import mformer_oracle as O  # noqa: E402

TASKS = (("taxa_L10", 1000), ("taxa_L20", 300), ("taxa_L30", 80), ("taxa_L40", 20))
cores = os.cpu_count()
torch.set_num_threads(cores)
cfg = get_default_config()
arch = load_config("/root/reference/configs/model/archs/mFormerV1/mFormerV1_sm.yaml")
cfg.MODEL = merge_configs(cfg.MODEL, arch.MODEL)
cfg.MODEL.IMG_SIZE = 224
cfg.MODEL.USE_FLASH_ATTN = False
cfg.MODEL.DROP_PATH_RATE = 0.0
cfg.TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS = False
cfg.DATA.TASK_KEYS_H5 = [t for t, _ in TASKS]
cfg.MODEL.CLASSIFICATION.HEADS = CN({t: {"TYPE": "Linear"} for t, _ in TASKS})
model = build_model(cfg, num_classes=dict(TASKS))
B = 8
g = torch.Generator().manual_seed(42)
x = torch.rand(B, 3, 224, 224, generator=g)
meta = torch.rand(B, 5, generator=g)
tg = {t: torch.randint(1, c, (B,), generator=g) for t, c in TASKS}
spec = O.Spec(heads=TASKS, drop_path_rate=0.0)
sd = {k: v.requires_grad_(True) for k, v in O.seeded_state_dict(O.param_shapes(spec), 1).items()}


def timed(fn, warm=3, n=10):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return B * n / (time.perf_counter() - t0)


def ref_fwd():
    with torch.no_grad():
        model(x, meta)


def ref_step():
    model.zero_grad(set_to_none=True)
    out = model(x, meta)
    sum(F.cross_entropy(out[t], tg[t]) for t, _ in TASKS).backward()


def or_fwd():
    with torch.no_grad():
        O.forward(sd, spec, x, meta)


def or_step():
    for v in sd.values():
        v.grad = None
    out = O.forward(sd, spec, x, meta)
    sum(F.cross_entropy(out[t], tg[t]) for t, _ in TASKS).backward()


res = {"where": "build container", "cores": cores, "batch": B, "protocol": "3 warm-up + 10 timed steps, fp32, 224x224, 4 Linear heads"}
model.eval()
res["reference_fwd_images_per_sec"] = round(timed(ref_fwd), 2)
model.train()
res["reference_fwd_bwd_images_per_sec"] = round(timed(ref_step), 2)
res["oracle_fwd_images_per_sec"] = round(timed(or_fwd), 2)
res["oracle_fwd_bwd_images_per_sec"] = round(timed(or_step), 2)
print(json.dumps(res))
with open(os.path.join(REPO, "profiles", "r03_reference_cpu_timing.json"), "w") as f:
    json.dump(res, f, indent=1)
