"""Minimal configuration objects for the mFormerV1 path.

The reference drives model construction from a yacs ``CfgNode`` (linnaeus/config.py).  This
package only needs the handful of keys mFormerV1 reads ("Config inputs
read"), so it accepts *any* object with attribute access and ``.get`` -- a real yacs CfgNode
from the reference works unchanged -- and ships ``ConfigNode`` + ``default_config()`` so the
path is usable without yacs or the reference installed.
"""
from __future__ import annotations

import copy
import os
from typing import Any

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


class ConfigNode(dict):
    """dict with attribute access, nested, with yacs-like get/clone/merge helpers."""

    def __init__(self, init: dict | None = None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = ConfigNode(v) if isinstance(v, dict) and not isinstance(v, ConfigNode) else v

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = ConfigNode(value) if isinstance(value, dict) and not isinstance(value, ConfigNode) else value

    def clone(self) -> "ConfigNode":
        return copy.deepcopy(self)

    def merge(self, other: dict) -> "ConfigNode":
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = ConfigNode(v) if isinstance(v, dict) else copy.deepcopy(v)
        return self


def default_config() -> ConfigNode:
    """Defaults of the keys on the mFormerV1 path (values as in the reference's
    linnaeus/config.py:268-306,412-470,560-566 and configs/model/archs/mFormerV1)."""
    return ConfigNode({
        "MODEL": {
            "TYPE": "mFormerV1", "NAME": "mFormerV1_sm", "IMG_SIZE": 384, "IN_CHANS": 3,
            "DROP_RATE": 0.0, "DROP_PATH_RATE": 0.2, "ATTN_DROP_RATE": 0.0, "LABEL_SMOOTHING": 0.1,
            "ONLY_LAST_CLS": False, "EXTRA_TOKEN_NUM": 4, "USE_FLASH_ATTN": False,
            "PRETRAINED": None, "PRETRAINED_SOURCE": None, "META_DIMS": [4, 3],
            "CONVNEXT_STAGES": {"DEPTHS": [3, 3, 9, 3], "DIMS": [96, 192, 384, 768], "LAYER_SCALE_INIT_VALUE": 1.0e-6},
            "ROPE_STAGES": {"DEPTHS": [5, 2], "DIMS": [384, 768], "NUM_HEADS": [6, 12], "MLP_RATIO": [4.0, 4.0],
                            "ROPE_THETA": 10000.0, "ROPE_MIXED": True},
            "CLASSIFICATION": {"HEADS": {}},
        },
        "DATA": {
            "TASK_KEYS_H5": [],
            "META": {"ACTIVE": True, "COMPONENTS": {
                "TEMPORAL": {"ENABLED": True, "DIM": 2, "IDX": 0},
                "SPATIAL": {"ENABLED": True, "DIM": 3, "IDX": 1},
                "ELEVATION": {"ENABLED": False, "DIM": 10, "IDX": 2},
            }},
        },
        "TRAIN": {"GRADIENT_CHECKPOINTING": {"ENABLED_NORMAL_STEPS": False}},
        "DEBUG": {"MODEL_BUILD": False},
    })


ARCHS = ("sm", "md", "lg", "xl")


def arch_config(arch: str = "sm", img_size: int = 224) -> ConfigNode:
    """default_config() merged with configs/mFormerV1_<arch>.yaml (our own data files with the
    reference's architecture hyper-parameters)."""
    if arch not in ARCHS:
        raise ValueError(f"unknown mFormerV1 arch '{arch}', expected one of {ARCHS}")
    cfg = default_config()
    with open(os.path.join(_HERE, "configs", f"mFormerV1_{arch}.yaml")) as f:
        cfg.merge(yaml.safe_load(f))
    cfg.MODEL.IMG_SIZE = img_size
    return cfg
