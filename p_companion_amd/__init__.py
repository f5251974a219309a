"""p_companion_amd -- MI355X-native (gfx950) training path for P-Companion's two
embedding-learning hot loops, behind the reference's own module surface.

Everything numerical runs in hand-written HIP kernels (csrc/, libpcompanion_hip.so,
C ABI in include/pcompanion_hip.h).  There is no CPU fallback: constructing the modules
on a machine without the built library or without a GPU raises.
"""
import torch  # noqa: F401  (first: see _lib.lib on the single-HIP-runtime rule)

from . import _lib  # noqa: F401
from ._lib import HipKernelError  # noqa: F401

__all__ = ["HipKernelError"]
