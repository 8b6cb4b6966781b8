"""linnaeus_amd -- MI355X-native mFormerV1 forward/backward path (HIP/gfx950 + RCCL)."""
__version__ = "0.1.0"
