"""Model/head registries and build_model() -- the drop-in seam of the path.

Mirrors the reference's plug-in interface for this path (same names, argument meaning and
error behaviour): linnaeus/models/model_factory.py:72-135 (decorator registries, overwrite
with a warning), :179-213 (create_model: ValueError on an unknown MODEL.TYPE) and
linnaeus/models/build.py:52-111 (build_model signature).
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, Optional, Type

import torch.nn as nn

logger = logging.getLogger("linnaeus_amd")

_model_registry: Dict[str, Type[nn.Module]] = {}
_head_registry: Dict[str, Type[nn.Module]] = {}


def _make_register(registry: Dict[str, Type[nn.Module]], kind: str) -> Callable[[str], Callable]:
    def register(name: str):
        def deco(cls):
            if name in registry:
                logger.warning("%s '%s' is already registered. Overwriting.", kind, name)
            registry[name] = cls
            return cls

        return deco

    return register


register_model = _make_register(_model_registry, "model")
register_head = _make_register(_head_registry, "classification head")


def create_head(name: str, **kwargs: Any) -> nn.Module:
    if name not in _head_registry:
        raise ValueError(f"classification head '{name}' is not registered. Available classification heads: {list(_head_registry)}")
    return _head_registry[name](**kwargs)


def create_model(config, **kwargs: Any) -> nn.Module:
    model_type = config.MODEL.TYPE
    if model_type not in _model_registry:
        raise ValueError(f"Unknown model type: {model_type}")
    return _model_registry[model_type](config, **kwargs)


def load_pretrained(config, model: nn.Module, strict: bool = False):
    """Stand-alone counterpart of the reference's single-source `load_pretrained` (utils/checkpoint.py:513-700) for a LOCAL
    checkpoint file: `torch.load` -> the state dict under "model" / "state_dict_ema" / "state_dict" (or the object itself) ->
    `module.` prefixes stripped -> keys matching the target model's `pretrained_ckpt_handling_metadata["drop_params"]` (regex:
    a pattern ending in "." as it stands, otherwise with a word boundary) and `["drop_buffers"]` (substring) dropped ->
    `load_state_dict(strict=False)`.  What needs the reference's tooling raises: bucket / hf:// paths, the MetaFormer key
    mapping and the stitched ConvNeXt + RoPE-ViT loading.  Under `install_into_linnaeus()` the reference's own
    `build_model` calls its own loader on this model (names and shapes are identical), so none of this is involved."""
    import os
    import re

    import torch

    M = config.MODEL
    if M.get("PRETRAINED_CONVNEXT", None) and M.get("PRETRAINED_ROPEVIT", None):
        raise NotImplementedError("stitched ConvNeXt + RoPE-ViT loading: use the reference's load_stitched_pretrained() (utils/checkpoint.py:216)")
    path = M.get("PRETRAINED", None)
    if not path:
        logger.warning("No pretrained checkpoint specified.")
        return None
    if M.get("PRETRAINED_SOURCE", None) == "metaformer":
        raise NotImplementedError("PRETRAINED_SOURCE 'metaformer': use the reference's map_metaformer_checkpoint() (utils/checkpoint.py)")
    if "://" in str(path) or not os.path.isfile(str(path)):
        raise FileNotFoundError(f"MODEL.PRETRAINED={path!r}: only local files are resolved here (bucket / hub paths: reference's resolve_checkpoint_path)")
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    raw = ckpt.get("model", ckpt.get("state_dict_ema", ckpt.get("state_dict", ckpt))) if isinstance(ckpt, dict) else ckpt
    if not raw:
        raise KeyError("Could not find model state dict in checkpoint.")
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in raw.items()}
    meta = getattr(model, "pretrained_ckpt_handling_metadata", {})
    for pattern in meta.get("drop_params", []):
        rx = pattern if pattern.endswith(".") else pattern + r"\b"
        for k in [k for k in sd if re.search(rx, k)]:
            del sd[k]
    for pattern in meta.get("drop_buffers", []):
        for k in [k for k in sd if pattern in k]:
            del sd[k]
    return model.load_state_dict(sd, strict=strict and meta.get("strict", False))


def build_model(config, num_classes: Optional[Dict[str, int]] = None, taxonomy_tree: Any = None) -> nn.Module:
    """Build a model from the final configuration (reference: models/build.py:52-111), including the pretrained-weight
    loading step at its end (:94-103) for local single-source checkpoints (`load_pretrained` above)."""
    model = create_model(config=config, num_classes=num_classes, taxonomy_tree=taxonomy_tree)
    if config.MODEL.get("PRETRAINED", None) or (config.MODEL.get("PRETRAINED_CONVNEXT", None) and config.MODEL.get("PRETRAINED_ROPEVIT", None)):
        load_pretrained(config, model, strict=False)
    return model


def install_into_linnaeus() -> bool:
    """Register this package's mFormerV1 under the reference's own registry, if `linnaeus` is
    importable: `linnaeus.models.build_model(cfg)` then constructs the HIP-backed model with no
    edits to the reference (model_factory.py:96-100 overwrites on re-registration)."""
    try:
        from linnaeus.models import model_factory as mf  # type: ignore
    except Exception:
        return False
    from .model import mFormerV1

    mf.register_model("mFormerV1")(mFormerV1)
    return True
