"""MI355X-native render path for ccelio/TileCodeRayTracer.

The product is the C-ABI library ``lib/libtcrt.so`` (include/rt_capi.h):
hand-written HIP kernels for gfx950 behind "render every pixel of a
rectangle".  This package is the thin Python side used by the tests and by
bench.py: ctypes bindings to that library (:mod:`.capi`), bindings to the C++
host model that mirrors the reference's Scene / SceneObject / Camera API
(:mod:`.host`), and a small convenience wrapper (:mod:`.renderer`).

There is no CPU rendering path anywhere in this package; if the HIP library is
missing or no GPU is present, rendering raises.
"""
from .capi import RtError, load_library, library_path  # noqa: F401
from .host import HostScene  # noqa: F401
from .renderer import Renderer  # noqa: F401

__all__ = ["RtError", "load_library", "library_path", "HostScene", "Renderer"]
