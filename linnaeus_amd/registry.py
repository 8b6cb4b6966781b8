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


def build_model(config, num_classes: Optional[Dict[str, int]] = None, taxonomy_tree: Any = None) -> nn.Module:
    """Build a model from the final configuration (reference: models/build.py:52-111).

    Pretrained-checkpoint loading (config.MODEL.PRETRAINED) belongs to the reference's
    checkpoint tooling, which is outside this path: a non-empty value is rejected loudly."""
    model = create_model(config=config, num_classes=num_classes, taxonomy_tree=taxonomy_tree)
    if config.MODEL.get("PRETRAINED", None):
        raise NotImplementedError(
            "MODEL.PRETRAINED: load the checkpoint with the reference's load_pretrained()/load_state_dict(); "
            "state_dict names and shapes are identical"
        )
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
