"""linnaeus_amd -- MI355X-native mFormerV1 forward/backward path (HIP/gfx950 + RCCL).

Drop-in for the reference's `linnaeus.models.build_model()` on the mFormerV1 path:

    from linnaeus_amd import build_model, arch_config
    model = build_model(arch_config("sm", 224), num_classes={...}).cuda()

or, with the reference installed, `linnaeus_amd.install_into_linnaeus()` re-registers
"mFormerV1" in the reference's own registry.
"""
__version__ = "0.1.0"

from .config import ConfigNode, arch_config, default_config  # noqa: F401
from .registry import build_model, create_model, install_into_linnaeus, register_head, register_model  # noqa: F401
from .model import mFormerV1  # noqa: F401
from . import autobatch, loss, optim, prefetch  # noqa: F401
