"""ctypes bindings for the drop-in boundary, include/rt_capi.h, and for the
speed-only options and diagnostics of include/rt_capi_tuning.h.

Struct layouts below must match the headers field for field.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_CAPACITY, RT_ERR_RCCL = range(6)
RT_KIND_SPHERE, RT_KIND_INFINITE_PLANE, RT_KIND_FINITE_PLANE = 0, 1, 2

F3 = C.c_float * 3


class RtObjectDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("is_light", C.c_int32), ("texture", C.c_int32),
        ("intensity", C.c_float),
        ("origin", F3), ("color", F3),
        ("diffuse", C.c_float), ("specular", C.c_float), ("reflective", C.c_float),
        ("radius", C.c_float), ("radius_squared", C.c_float),
        ("plane_origin", F3),
        ("normal", F3), ("vertical", F3), ("horizontal", F3), ("reverse_normal", F3),
        ("v_distance", C.c_float), ("h_distance", C.c_float),
        ("distance_to_origin", C.c_float),
    ]


class RtTextureDesc(C.Structure):
    _fields_ = [("light", F3), ("dark", F3), ("width", C.c_float), ("height", C.c_float)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("n_objects", C.c_int32), ("objects", C.POINTER(RtObjectDesc)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(RtTextureDesc)),
        ("shadow_begin", C.c_int32), ("shadow_end", C.c_int32),
        ("null_color", F3),
    ]


class RtCameraDesc(C.Structure):
    _fields_ = [
        ("screen_width", C.c_float), ("screen_height", C.c_float),
        ("screen_halfwidth", C.c_float), ("screen_halfheight", C.c_float),
        ("screen_origin", F3), ("vector_horizontal", F3), ("vector_vertical", F3),
        ("eye_origin", F3),
    ]


class RtTiming(C.Structure):
    _fields_ = [
        ("last_kernel_ms", C.c_double), ("sum_kernel_ms", C.c_double),
        ("launches", C.c_uint64),
        ("last_upload_ms", C.c_double), ("last_download_ms", C.c_double),
    ]


class RtLaunchInfo(C.Structure):
    _fields_ = [
        ("block_threads", C.c_int32), ("lds_bytes", C.c_int32), ("scene_lds_bytes", C.c_int32),
        ("grid_blocks", C.c_int32), ("tile_x", C.c_int32), ("tile_z", C.c_int32),
        ("kernel", C.c_char * 48),
    ]


RT_MULTI_MAX_GPUS = 16


class RtMultiInfo(C.Structure):
    _fields_ = [
        ("ngpu", C.c_int32), ("chunks", C.c_int32), ("balanced", C.c_int32), ("transport", C.c_int32),
        ("bounds", C.c_int32 * (RT_MULTI_MAX_GPUS + 1)),
        ("kernel_ms", C.c_double * RT_MULTI_MAX_GPUS),
        ("frame_ms", C.c_double),
        ("measured_kernel_ms", C.c_double * RT_MULTI_MAX_GPUS),
        ("measured_gather_ms", C.c_double),
        ("trial_image_ok", C.c_int32 * 2),
        ("trial_frame_ms", C.c_double * 2),
    ]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_capi error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def library_path():
    """lib/libtcrt.so next to this package, unless TCRT_LIBRARY names another build of it."""
    return os.environ.get("TCRT_LIBRARY") or os.path.join(LIB_DIR, "libtcrt.so")


def load_library():
    """Load lib/libtcrt.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `make -C tilecoderaytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i = C.c_void_p, C.c_int
    lib.rt_capi_version.restype = i
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_device_count.argtypes = [C.POINTER(i)]
    lib.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), i, C.POINTER(vp)]
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_render.argtypes = [vp, C.POINTER(RtCameraDesc), i, i, i, i, i, vp]
    lib.rt_render_device.argtypes = [vp, C.POINTER(RtCameraDesc), i, i, i, i, i, vp, vp]
    lib.rt_render_multi.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(RtCameraDesc), i, i, i, i, vp]
    lib.rt_strip_bounds.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.rt_strip_bounds.restype = i
    lib.rt_chunk_bounds.argtypes = [i, i, i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.rt_multi_create.argtypes = [C.POINTER(RtSceneDesc), i, C.POINTER(vp)]
    lib.rt_multi_render.argtypes = [vp, C.POINTER(RtCameraDesc), i, i, i, i, vp]
    lib.rt_multi_set_option.argtypes = [vp, C.c_char_p, i]
    lib.rt_multi_set_bounds.argtypes = [vp, i, C.POINTER(i), i]
    lib.rt_multi_get_info.argtypes = [vp, C.POINTER(RtMultiInfo)]
    lib.rt_balance_strips.argtypes = [i, i, C.POINTER(i), C.POINTER(C.c_double), C.c_double, i, C.POINTER(i)]
    lib.rt_suggest_chunks.argtypes = [C.c_double, C.c_double, i]
    lib.rt_capi_tuning_version.restype = i
    lib.rt_shared_image_create.argtypes = [i, C.c_uint64, C.POINTER(vp), C.c_char_p]
    lib.rt_shared_image_open.argtypes = [i, C.c_char_p, C.POINTER(vp)]
    lib.rt_shared_image_close.argtypes = [i, vp]
    lib.rt_shared_image_destroy.argtypes = [i, vp]
    lib.rt_multi_destroy.argtypes = [vp]
    lib.rt_render_stats.argtypes = [vp, C.POINTER(RtCameraDesc), i, i, i, i, i, vp, C.POINTER(C.c_uint64), i, vp, i]
    lib.rt_learn_tile_order.argtypes = [vp, C.POINTER(RtCameraDesc), i, i, i, i, i]
    lib.rt_get_timing.argtypes = [vp, C.POINTER(RtTiming)]
    lib.rt_get_timeline.argtypes = [vp, vp, i]
    lib.rt_get_timeline.restype = i
    lib.rt_reset_timing.argtypes = [vp]
    lib.rt_get_launch_info.argtypes = [vp, C.POINTER(RtLaunchInfo)]
    lib.rt_set_option.argtypes = [vp, C.c_char_p, i]
    for name in ("rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_render",
                 "rt_render_device", "rt_render_multi", "rt_render_stats", "rt_learn_tile_order", "rt_get_timing", "rt_reset_timing",
                 "rt_get_launch_info", "rt_set_option", "rt_chunk_bounds", "rt_multi_create", "rt_multi_render",
                 "rt_multi_set_option", "rt_multi_destroy", "rt_multi_set_bounds", "rt_multi_get_info", "rt_balance_strips",
                 "rt_suggest_chunks", "rt_shared_image_create", "rt_shared_image_open", "rt_shared_image_close",
                 "rt_shared_image_destroy"):
        getattr(lib, name).restype = i
    _lib = lib
    return lib


def check(rc):
    if rc != RT_OK:
        msg = load_library().rt_last_error()
        raise RtError(rc, msg.decode("utf-8", "replace") if msg else "")
