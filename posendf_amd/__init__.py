"""posendf_amd -- MI355X-native Pose-NDF distance / projection engine (gfx950 HIP kernels).

`from posendf_amd import PoseNDF` is the drop-in for `from model.posendf import PoseNDF` of the reference.
"""
from . import synth  # noqa: F401
from .config import amass_config, load_config  # noqa: F401


def __getattr__(name):   # torch is imported lazily so that numpy-only users (oracle, packer tests) stay light
    if name in ("PoseNDF", "gradient"):
        from . import facade
        return getattr(facade, name)
    if name == "BodyModel":
        from .body_model import BodyModel
        return BodyModel
    raise AttributeError(name)


__all__ = ["PoseNDF", "gradient", "BodyModel", "amass_config", "load_config", "synth"]
