"""Convenience wrapper over the C ABI for tests and bench.py."""
import ctypes as C

import numpy as np

from . import capi


class Renderer:
    """Owns an ``rt_scene`` (device tables for one HostScene on one GPU)."""

    def __init__(self, host_scene, device=0):
        self._lib = capi.load_library()
        self._host = host_scene          # keeps the desc arrays alive
        self._scene = C.c_void_p()
        capi.check(self._lib.rt_scene_create(host_scene.desc, device, C.byref(self._scene)))
        self._cam = host_scene.camera

    @classmethod
    def from_desc(cls, desc, camera, device=0, keepalive=None):
        """Build from raw RtSceneDesc / RtCameraDesc (tests with hand-made tables)."""
        self = cls.__new__(cls)
        self._lib = capi.load_library()
        self._host = keepalive
        self._scene = C.c_void_p()
        capi.check(self._lib.rt_scene_create(C.byref(desc), device, C.byref(self._scene)))
        self._cam = C.pointer(camera)
        return self

    def close(self):
        if getattr(self, "_scene", None):
            self._lib.rt_scene_destroy(self._scene)
            self._scene = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        capi.check(self._lib.rt_set_option(self._scene, key.encode(), int(value)))

    def render(self, W, H, max_depth, x0=0, x1=None):
        """Columns [x0, x1) of a W x H image -> float32 array (x1-x0, H, 3)."""
        x1 = W if x1 is None else x1
        out = np.empty((max(x1 - x0, 0), H, 3), dtype=np.float32)
        capi.check(self._lib.rt_render(self._scene, self._cam, W, H, x0, x1, max_depth,
                                       out.ctypes.data))
        return out

    def render_device(self, W, H, max_depth, x0, x1, device_ptr, stream=0):
        """Enqueue a render into device memory on a HIP stream (no sync)."""
        capi.check(self._lib.rt_render_device(self._scene, self._cam, W, H, x0, x1, max_depth,
                                              C.c_void_p(device_ptr), C.c_void_p(stream)))

    STAT_NAMES = ("nearest_rays", "shadow_rays", "wave_nearest_scans", "wave_shadow_scans",
                  "wave_sphere_tests", "wave_plane_tests", "wave_box_tests", "lane_sphere_tests",
                  "cycles_nearest", "cycles_shadow", "cycles_tile",
                  "cycles_winner", "cycles_lights", "cycles_reflect",
                  "shadow_candidates", "shadow_leaves_union", "shadow_leaves_maxlane",
                  "nearest_scans_1_16", "nearest_scans_17_32", "nearest_scans_33_48", "nearest_scans_49_64",
                  "nearest_scans_unculled", "nearest_unculled_box_tests", "nearest_unculled_sphere_tests", "nearest_sphere_tests")

    def learn_tile_order(self, W, H, max_depth, x0=0, x1=None):
        """rt_learn_tile_order: one frame of the counting build on this launch shape; later renders of the same shape hand their
        tile rows out most expensive first (speed only; set_option("learned_order", 0) forgets it)."""
        capi.check(self._lib.rt_learn_tile_order(self._scene, self._cam, W, H, x0, W if x1 is None else x1, max_depth))

    def render_stats(self, W, H, max_depth, x0=0, x1=None, wave_cycles=False):
        """Counting build: returns (image, {counter: value}[, per wavefront tile (tiles_z, tiles_x, 6) = cycles, sphere tests, box tests, scans, start, end (100 MHz)])."""
        x1 = W if x1 is None else x1
        out = np.empty((max(x1 - x0, 0), H, 3), dtype=np.float32)
        st = (C.c_uint64 * len(self.STAT_NAMES))()
        li = self.launch_info()
        tz = li.tile_z or 4
        tx = 64 // tz
        tiles = ((H + tz - 1) // tz, (x1 - x0 + tx - 1) // tx)      # (tile rows, tile columns), row-major
        cyc = np.zeros(tiles + (6,), dtype=np.uint64)
        capi.check(self._lib.rt_render_stats(self._scene, self._cam, W, H, x0, x1, max_depth,
                                             out.ctypes.data, st, len(self.STAT_NAMES),
                                             cyc.ctypes.data if wave_cycles else None, cyc.size if wave_cycles else 0))
        stats = dict(zip(self.STAT_NAMES, [int(v) for v in st]))
        return (out, stats, cyc) if wave_cycles else (out, stats)

    def timeline(self, x0, x1, H):
        """Per wavefront tile of the last launch (option "timeline" = 1): array (tile rows, tile columns, 4) =
        start, end (100 MHz clock), workgroup * 16 + wavefront, rendered-as-a-HEAVY-tile."""
        li = self.launch_info()
        tiles = ((H + li.tile_z - 1) // li.tile_z, (x1 - x0 + li.tile_x - 1) // li.tile_x)
        rec = np.zeros(tiles + (4,), dtype=np.uint64)
        capi.check(self._lib.rt_get_timeline(self._scene, rec.ctypes.data, rec.size))
        return rec

    def timing(self):
        t = capi.RtTiming()
        capi.check(self._lib.rt_get_timing(self._scene, C.byref(t)))
        return t

    def reset_timing(self):
        capi.check(self._lib.rt_reset_timing(self._scene))

    def launch_info(self):
        li = capi.RtLaunchInfo()
        capi.check(self._lib.rt_get_launch_info(self._scene, C.byref(li)))
        return li
