"""Bindings to lib/libtcrt_host.so: the C++ host model that mirrors the
reference's Scene / SceneObject / Camera API (csrc/host/celio_model.hpp).

A :class:`HostScene` is a reference ``Scene`` plus a ``Camera``; it is built
with the same calls a user of the reference makes (``add_sphere`` = ``new
SceneSphere`` + ``addObject`` ...), and ``desc`` / ``camera`` give the
flattened plain-old-data views that cross the C ABI.  Nothing here renders.
"""
import ctypes as C
import os

from .capi import LIB_DIR, RtCameraDesc, RtSceneDesc, F3

_hlib = None


def host_library_path():
    return os.path.join(LIB_DIR, "libtcrt_host.so")


def load_host_library():
    global _hlib
    if _hlib is not None:
        return _hlib
    path = host_library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `make -C tilecoderaytracer_amd/csrc`")
    lib = C.CDLL(path)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    pf = C.POINTER(C.c_float)
    lib.rth_scene_new.argtypes = [C.POINTER(vp)]
    lib.rth_scene_builtin.argtypes = [C.POINTER(vp)]
    lib.rth_scene_two_mirrors.argtypes = [C.POINTER(vp)]
    lib.rth_scene_grid.argtypes = [i, i, C.POINTER(vp)]
    lib.rth_scene_free.argtypes = [vp]
    lib.rth_scene_free.restype = None
    lib.rth_add_sphere.argtypes = [vp, pf, f]
    lib.rth_add_infinite_plane.argtypes = [vp, pf, pf, pf]
    lib.rth_add_finite_plane_corners.argtypes = [vp, pf, pf, pf]
    lib.rth_add_finite_plane_axes.argtypes = [vp, pf, pf, pf, f, f]
    lib.rth_object_count.argtypes = [vp]
    lib.rth_set_color.argtypes = [vp, i, pf]
    lib.rth_set_diffuse.argtypes = [vp, i, f]
    lib.rth_set_specular.argtypes = [vp, i, f]
    lib.rth_set_reflective.argtypes = [vp, i, f]
    lib.rth_set_checkerboard.argtypes = [vp, i, pf, pf, f, f]
    lib.rth_set_light.argtypes = [vp, i]
    lib.rth_set_intensity.argtypes = [vp, i, f]
    lib.rth_set_object_indices.argtypes = [vp, i, i]
    lib.rth_camera_two_mirrors.argtypes = [vp]
    lib.rth_camera_eye_ray.argtypes = [vp, f, f, pf, pf]
    lib.rth_scene_desc.argtypes = [vp]
    lib.rth_scene_desc.restype = C.POINTER(RtSceneDesc)
    lib.rth_camera_desc.argtypes = [vp]
    lib.rth_camera_desc.restype = C.POINTER(RtCameraDesc)
    lib.rth_write_screen_txt.argtypes = [C.c_char_p, i, i, vp, C.c_double, C.c_double]
    lib.rth_write_screen_txt_cores.argtypes = [C.c_char_p, i, i, vp, C.c_double, C.c_double, i]
    _hlib = lib
    return lib


def _v(p):
    return F3(*[float(x) for x in p])


class HostScene:
    """A reference ``Scene`` + ``Camera`` living in the C++ host model."""

    def __init__(self, handle):
        self._lib = load_host_library()
        self._h = handle

    # -- whole scenes ---------------------------------------------------
    @classmethod
    def _make(cls, fn, *args):
        lib = load_host_library()
        h = C.c_void_p()
        if fn(lib)(*args, C.byref(h)) != 0 or not h:
            raise RuntimeError("host scene construction failed")
        return cls(h)

    @classmethod
    def empty(cls):
        return cls._make(lambda l: l.rth_scene_new)

    @classmethod
    def builtin(cls):
        """``Scene::initialize()`` + default ``Camera()`` (the museum)."""
        return cls._make(lambda l: l.rth_scene_builtin)

    @classmethod
    def two_mirrors(cls):
        """``Scene::initializeTwoMirrors(&camera)``."""
        return cls._make(lambda l: l.rth_scene_two_mirrors)

    @classmethod
    def grid(cls, n, shadows=True):
        """Synthetic grid-n scene; camera = setSceneTwoMirrors()."""
        return cls._make(lambda l: l.rth_scene_grid, int(n), 1 if shadows else 0)

    @classmethod
    def named(cls, name):
        """'builtin' | 'twomirrors' | 'grid<N>' | 'grid<N>-noshadow'."""
        if name == "builtin":
            return cls.builtin()
        if name == "twomirrors":
            return cls.two_mirrors()
        if name.startswith("grid"):
            body = name[4:]
            shadows = not body.endswith("-noshadow")
            if not shadows:
                body = body[: -len("-noshadow")]
            return cls.grid(int(body), shadows)
        raise ValueError(f"unknown scene {name!r}")

    def close(self):
        if self._h:
            self._lib.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference-style building ---------------------------------------
    def add_sphere(self, origin, radius):
        return self._lib.rth_add_sphere(self._h, _v(origin), radius)

    def add_infinite_plane(self, o, n, h):
        return self._lib.rth_add_infinite_plane(self._h, _v(o), _v(n), _v(h))

    def add_finite_plane_corners(self, o, vcorner, hcorner):
        return self._lib.rth_add_finite_plane_corners(self._h, _v(o), _v(vcorner), _v(hcorner))

    def add_finite_plane_axes(self, o, n, h, v_dist, h_dist):
        return self._lib.rth_add_finite_plane_axes(self._h, _v(o), _v(n), _v(h), v_dist, h_dist)

    @property
    def object_count(self):
        return self._lib.rth_object_count(self._h)

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError("host model call failed (bad index?)")

    def set_color(self, idx, rgb): self._ok(self._lib.rth_set_color(self._h, idx, _v(rgb)))
    def set_diffuse(self, idx, f): self._ok(self._lib.rth_set_diffuse(self._h, idx, f))
    def set_specular(self, idx, f): self._ok(self._lib.rth_set_specular(self._h, idx, f))
    def set_reflective(self, idx, f): self._ok(self._lib.rth_set_reflective(self._h, idx, f))
    def set_checkerboard(self, idx, light, dark, w, h):
        self._ok(self._lib.rth_set_checkerboard(self._h, idx, _v(light), _v(dark), w, h))
    def set_light(self, idx): self._ok(self._lib.rth_set_light(self._h, idx))
    def set_intensity(self, idx, f): self._ok(self._lib.rth_set_intensity(self._h, idx, f))
    def set_object_indices(self, my_rank, group_size):
        self._ok(self._lib.rth_set_object_indices(self._h, my_rank, group_size))
    def camera_two_mirrors(self): self._ok(self._lib.rth_camera_two_mirrors(self._h))

    def eye_ray(self, dx, dy):
        o, d = F3(), F3()
        self._ok(self._lib.rth_camera_eye_ray(self._h, dx, dy, o, d))
        return tuple(o), tuple(d)

    # -- flattened views (valid while self is alive and unmodified) -----
    @property
    def desc(self):
        return self._lib.rth_scene_desc(self._h)

    @property
    def camera(self):
        return self._lib.rth_camera_desc(self._h)


def write_screen_txt(path, rgb, run_time_s=0.0, us_per_pixel=0.0, n_cores=1):
    """Write ``raytracer_screen.txt`` for a (W, H, 3) float32 array; n_cores = GPUs that rendered it
    (CORE_NUM and the partition label of the header, src/RayTracer.cpp:2037-2058)."""
    import numpy as np
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    W, H = a.shape[0], a.shape[1]
    rc = load_host_library().rth_write_screen_txt_cores(os.fsencode(path), W, H, a.ctypes.data,
                                                        run_time_s, us_per_pixel, int(n_cores))
    if rc != 0:
        raise OSError(f"could not write {path}")
