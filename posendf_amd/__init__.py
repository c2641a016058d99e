"""posendf_amd -- MI355X-native Pose-NDF distance / projection engine (gfx950 HIP kernels)."""
from . import synth  # noqa: F401

__all__ = ["synth"]
