"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product never does.
"""
import ctypes as C
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "oracle", "liboracle.so")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class OrcObject(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("origin", Vec3), ("is_light", C.c_int), ("intensity", C.c_float),
        ("color", Vec3), ("diffuse", C.c_float), ("specular", C.c_float), ("reflective", C.c_float),
        ("has_texture", C.c_int), ("tex_light", Vec3), ("tex_dark", Vec3),
        ("tex_width", C.c_float), ("tex_height", C.c_float),
        ("radius", C.c_float), ("radius_squared", C.c_float),
        ("plane_origin", Vec3), ("normal", Vec3), ("vertical", Vec3), ("horizontal", Vec3),
        ("reverse_normal", Vec3),
        ("v_distance", C.c_float), ("h_distance", C.c_float), ("distance_to_origin", C.c_float),
    ]


class OrcCamera(C.Structure):
    _fields_ = [
        ("screen_width", C.c_float), ("screen_height", C.c_float),
        ("screen_halfwidth", C.c_float), ("screen_halfheight", C.c_float),
        ("screen_origin", Vec3), ("vector_outwards", Vec3), ("vector_vertical", Vec3),
        ("vector_horizontal", Vec3), ("eye_distance", C.c_float), ("eye_origin", Vec3),
    ]


class OrcCounters(C.Structure):
    _fields_ = [("nearest_rays", C.c_ulonglong), ("shadow_rays", C.c_ulonglong),
                ("collision_tests", C.c_ulonglong)]


def _load():
    if not os.path.exists(_PATH):
        raise RuntimeError(f"{_PATH} missing: run `make -C oracle`")
    L = C.CDLL(_PATH)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.orc_scene_new.restype = vp
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_free.restype = None
    L.orc_scene_object_count.argtypes = [vp]
    L.orc_scene_get_object.argtypes = [vp, i, C.POINTER(OrcObject)]
    L.orc_scene_shadow_range.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.orc_add_sphere.argtypes = [vp, Vec3, f]
    L.orc_add_infinite_plane.argtypes = [vp, Vec3, Vec3, Vec3]
    L.orc_add_finite_plane_corners.argtypes = [vp, Vec3, Vec3, Vec3]
    L.orc_add_finite_plane_axes.argtypes = [vp, Vec3, Vec3, Vec3, f, f]
    L.orc_set_color.argtypes = [vp, i, Vec3]
    L.orc_set_diffuse.argtypes = [vp, i, f]
    L.orc_set_specular.argtypes = [vp, i, f]
    L.orc_set_reflective.argtypes = [vp, i, f]
    L.orc_set_checkerboard.argtypes = [vp, i, Vec3, Vec3, f, f]
    L.orc_set_light.argtypes = [vp, i]
    L.orc_set_intensity.argtypes = [vp, i, f]
    L.orc_set_object_indices.argtypes = [vp, i, i]
    L.orc_set_object_indices.restype = None
    L.orc_scene_initialize.argtypes = [vp]
    L.orc_scene_initialize_two_mirrors.argtypes = [vp, C.POINTER(OrcCamera)]
    L.orc_scene_grid.argtypes = [vp, i, i]
    L.orc_camera_default.argtypes = [C.POINTER(OrcCamera)]
    L.orc_camera_default.restype = None
    L.orc_camera_two_mirrors.argtypes = [C.POINTER(OrcCamera)]
    L.orc_camera_two_mirrors.restype = None
    L.orc_camera_eye_ray.argtypes = [C.POINTER(OrcCamera), f, f, C.POINTER(Vec3), C.POINTER(Vec3)]
    L.orc_camera_eye_ray.restype = None
    L.orc_render.argtypes = [vp, C.POINTER(OrcCamera), i, i, i, i, i, vp]
    L.orc_render_static_partition.argtypes = [vp, C.POINTER(OrcCamera), i, i, i, C.POINTER(i), i, i, i, C.POINTER(i), vp,
                                              C.POINTER(C.c_double)]
    L.orc_get_counters.argtypes = [C.POINTER(OrcCounters)]
    L.orc_get_counters.restype = None
    L.orc_write_screen_txt.argtypes = [C.c_char_p, i, i, vp, C.c_double, C.c_double]
    return L


LIB = _load()


def vec(p):
    return Vec3(float(p[0]), float(p[1]), float(p[2]))


class OracleScene:
    """An oracle scene + camera, built with the same verbs as HostScene."""

    def __init__(self):
        self.h = LIB.orc_scene_new()
        self.cam = OrcCamera()
        LIB.orc_camera_default(C.byref(self.cam))

    def __del__(self):
        if getattr(self, "h", None):
            LIB.orc_scene_free(self.h)
            self.h = None

    @classmethod
    def builtin(cls):
        s = cls()
        assert LIB.orc_scene_initialize(s.h) == 0
        return s

    @classmethod
    def two_mirrors(cls):
        s = cls()
        assert LIB.orc_scene_initialize_two_mirrors(s.h, C.byref(s.cam)) == 0
        return s

    @classmethod
    def grid(cls, n, shadows=True):
        s = cls()
        assert LIB.orc_scene_grid(s.h, n, 1 if shadows else 0) == 0
        LIB.orc_camera_two_mirrors(C.byref(s.cam))
        return s

    @classmethod
    def named(cls, name):
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
        raise ValueError(name)

    # building verbs
    def add_sphere(self, o, r): return LIB.orc_add_sphere(self.h, vec(o), r)
    def add_infinite_plane(self, o, n, h): return LIB.orc_add_infinite_plane(self.h, vec(o), vec(n), vec(h))
    def add_finite_plane_corners(self, o, vc, hc):
        return LIB.orc_add_finite_plane_corners(self.h, vec(o), vec(vc), vec(hc))
    def add_finite_plane_axes(self, o, n, h, vd, hd):
        return LIB.orc_add_finite_plane_axes(self.h, vec(o), vec(n), vec(h), vd, hd)
    def set_color(self, i, c): assert LIB.orc_set_color(self.h, i, vec(c)) == 0
    def set_diffuse(self, i, f): assert LIB.orc_set_diffuse(self.h, i, f) == 0
    def set_specular(self, i, f): assert LIB.orc_set_specular(self.h, i, f) == 0
    def set_reflective(self, i, f): assert LIB.orc_set_reflective(self.h, i, f) == 0
    def set_checkerboard(self, i, l, d, w, h): assert LIB.orc_set_checkerboard(self.h, i, vec(l), vec(d), w, h) == 0
    def set_light(self, i): assert LIB.orc_set_light(self.h, i) == 0
    def set_intensity(self, i, f): assert LIB.orc_set_intensity(self.h, i, f) == 0
    def set_object_indices(self, rank, size): LIB.orc_set_object_indices(self.h, rank, size)
    def camera_two_mirrors(self): LIB.orc_camera_two_mirrors(C.byref(self.cam))

    @property
    def object_count(self):
        return LIB.orc_scene_object_count(self.h)

    def get_object(self, i):
        o = OrcObject()
        assert LIB.orc_scene_get_object(self.h, i, C.byref(o)) == 0
        return o

    def shadow_range(self):
        b, e = C.c_int(), C.c_int()
        LIB.orc_scene_shadow_range(self.h, C.byref(b), C.byref(e))
        return b.value, e.value

    def eye_ray(self, dx, dy):
        o, d = Vec3(), Vec3()
        LIB.orc_camera_eye_ray(C.byref(self.cam), dx, dy, C.byref(o), C.byref(d))
        return o.tuple(), d.tuple()

    def render(self, W, H, max_depth, x0=0, x1=None):
        x1 = W if x1 is None else x1
        out = np.empty((max(x1 - x0, 0), H, 3), dtype=np.float32)
        rc = LIB.orc_render(self.h, C.byref(self.cam), W, H, x0, x1, max_depth, out.ctypes.data)
        assert rc == 0
        return out

    @staticmethod
    def counters():
        c = OrcCounters()
        LIB.orc_get_counters(C.byref(c))
        return c


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


def write_screen_txt(path, rgb, run_time_s=0.0, us_per_pixel=0.0):
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    assert LIB.orc_write_screen_txt(os.fsencode(path), a.shape[0], a.shape[1], a.ctypes.data,
                                    run_time_s, us_per_pixel) == 0


# SHA-256 digests of packed fp32 framebuffers recorded in SURVEY.md Appendix D
# from the survey's scratch build of the reference.  The grid scenes there were
# assembled with addObject() alone, i.e. shadow scan range [0, 0) ("-noshadow").
SURVEY_PINS = {
    "b64d4":    ("builtin",          64,  64, 4, "8112d69522905d6d09c7d35704e5d9cfe880d4b4dd89c84a6c7c46d7290567cf"),
    "g32_64d4": ("grid32-noshadow",  64,  64, 4, "c3f0ba632a02d04176bc3024d713078d4c8b769f4543c7ac0135c38ee6146939"),
    "g16_64d8": ("grid16-noshadow",  64,  64, 8, "8521ac71df13a7e4cbeb1021f34dbc2a14619ea5e5cf43ea8b08c3fae0c65b2b"),
    "b256d4":   ("builtin",         256, 256, 4, "3041e286e437c92c5ed21c76022d23aa3529dd5f2220eaf83566b6494e81caa1"),
    "g32d4":    ("grid32-noshadow", 256, 256, 4, "5578e966be2ca991118eef2d7fe57560e18482dee957d5aa65fb75e6b65f7a50"),
    "g16d8":    ("grid16-noshadow", 256, 256, 8, "a040ba188493a1969a8aa83e8b84fac3c48feb9f7bb82f784c249b7ea5e1a815"),
    "b512d3":   ("builtin",         512, 512, 3, "dfc930b0f7c2268da9218b7e9322b4df5823baebc46540a466b30d813ff87ecd"),
}
