"""MI355X-native TensoRF-VM renderer for joint pose + radiance-field training.

The hot path (ray generation from se(3) pose parameters, sampling, VM-factor interpolation, the
appearance MLP, transmittance compositing, separable factor blur -- forward and backward) runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/jt_render.h.  Importing the package
loads joint_tensorf_amd/lib/libjt_render.so and fails loudly when it is missing.
"""
from . import _lib  # noqa: F401  (raises ImportError if the HIP library is not built)
from . import ops  # noqa: F401
from .tensorf_repr import BAT_VMSplit, TVLoss  # noqa: F401

__all__ = ["ops", "BAT_VMSplit", "TVLoss"]
