"""MI355X-native conditional sampling of VAEs: LAIR and AC-MWG on hand-written HIP kernels.

Host-side mirror of the sampler interfaces of vsimkus/vae-conditional-sampling
(``irwg.sampling``, ``irwg.models``) on top of ``libcsvae.so`` (include/csvae.h).
All compute on the sampling path runs in that library; there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
