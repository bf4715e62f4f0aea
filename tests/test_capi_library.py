"""The C-ABI library loads, exports every symbol include/rt_capi.h declares,
validates descriptions, and -- on a machine without a GPU -- refuses to render
instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from tilecoderaytracer_amd import HostScene, RtError, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="rt_capi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int|const char \*)\s*(rt_\w+)\s*\(", text, flags=re.M)))


DROP_IN = ["rt_balance_strips", "rt_capi_version", "rt_chunk_bounds", "rt_device_count", "rt_get_timing", "rt_last_error",
           "rt_multi_create", "rt_multi_destroy", "rt_multi_get_info", "rt_multi_render", "rt_multi_set_bounds",
           "rt_render", "rt_render_device", "rt_render_multi", "rt_reset_timing", "rt_scene_create", "rt_scene_destroy",
           "rt_shared_image_close", "rt_shared_image_create", "rt_shared_image_destroy", "rt_shared_image_open",
           "rt_strip_bounds", "rt_suggest_chunks"]
TUNING = ["rt_capi_tuning_version", "rt_get_launch_info", "rt_get_timeline", "rt_learn_tile_order", "rt_multi_set_option",
          "rt_render_stats", "rt_set_option"]


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for n in ("rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_device", "rt_render_multi",
              "rt_last_error", "rt_get_timing"):
        assert n in names


def test_the_two_headers_split_the_abi():
    """include/rt_capi.h is the drop-in surface (scenes, renders, the multi-GPU partition, timing, errors) and nothing
    else; options, the counting build, diagnostics and the calibration call are include/rt_capi_tuning.h, which has its
    own version number.  INTEGRATION.md sections 1-2 (what a maintainer of the reference adds) use the first only."""
    assert declared_functions("rt_capi.h") == DROP_IN
    assert declared_functions("rt_capi_tuning.h") == TUNING
    drop_in_text = open(os.path.join(ROOT, "include", "rt_capi.h")).read()
    assert "rt_set_option(" not in re.sub(r"/\*.*?\*/", "", drop_in_text, flags=re.S)
    assert '#include "rt_capi.h"' in open(os.path.join(ROOT, "include", "rt_capi_tuning.h")).read()
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    first_two = integration[integration.index("## 1."):integration.index("## 3.")]
    assert "rt_capi_tuning.h" not in first_two
    for name in TUNING:
        assert name not in first_two, name


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in declared_functions("rt_capi.h") + declared_functions("rt_capi_tuning.h"):
        assert getattr(lib, name) is not None, name
    assert lib.rt_capi_version() == 4 and lib.rt_capi_tuning_version() == 1
    for header, macro, fn in (("rt_capi.h", "RT_CAPI_VERSION", lib.rt_capi_version),
                              ("rt_capi_tuning.h", "RT_CAPI_TUNING_VERSION", lib.rt_capi_tuning_version)):
        text = open(os.path.join(ROOT, "include", header)).read()
        assert int(re.search(r"#define %s (\d+)" % macro, text).group(1)) == fn()


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers compile as C99 with warnings as errors (a maintainer of the reference, or any FFI
    generator, includes them from C)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "rt_capi.h"\n#include "rt_capi_tuning.h"\n'
                   "int main(void) { unsigned char h[RT_SHARED_HANDLE_BYTES]; rt_multi_info i; (void)h; (void)i;\n"
                   "  return (RT_MULTI_TRANSPORT_DIRECT == 2 && RT_CAPI_VERSION == 4) ? 0 : 1; }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_struct_sizes_match_the_header():
    # field-for-field mirrors; sizes as laid out by the C compiler
    assert C.sizeof(capi.RtObjectDesc) == 4 * (4 + 3 + 3 + 3 + 2 + 3 + 12 + 2 + 1)
    assert C.sizeof(capi.RtTextureDesc) == 32
    assert C.sizeof(capi.RtCameraDesc) == 64
    assert C.sizeof(capi.RtTiming) == 40
    assert C.sizeof(capi.RtLaunchInfo) == 24 + 48      # six int32 + kernel[48]
    assert C.sizeof(capi.RtMultiInfo) == 4 * 4 + 4 * 17 + 4 + 8 * 16 + 8 + 8 * 16 + 8 + 4 * 2 + 8 * 2      # (4 bytes of padding before the first doubles)


def _create(desc):
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.rt_scene_create(C.byref(desc), 0, C.byref(h))
    if h:
        lib.rt_scene_destroy(h)
    return rc, lib.rt_last_error().decode()


def _copy_desc(src):
    d = capi.RtSceneDesc()
    C.memmove(C.byref(d), src, C.sizeof(d))
    return d


def test_invalid_descriptions_are_rejected_before_touching_a_device():
    host = HostScene.builtin()
    d = _copy_desc(host.desc)
    d.n_objects = -1
    assert _create(d)[0] == capi.RT_ERR_INVALID
    d = _copy_desc(host.desc)
    d.shadow_end = 33
    rc, msg = _create(d)
    assert rc == capi.RT_ERR_INVALID and "shadow" in msg
    d = _copy_desc(host.desc)
    objs = (capi.RtObjectDesc * 32)(*[d.objects[i] for i in range(32)])
    objs[5].kind = 7
    d.objects = objs
    assert _create(d)[0] == capi.RT_ERR_INVALID
    objs[5].kind = 0
    objs[9].texture = 3
    assert _create(d)[0] == capi.RT_ERR_INVALID
    lib = capi.load_library()
    assert lib.rt_scene_create(None, 0, None) == capi.RT_ERR_INVALID
    assert lib.rt_render(None, None, 4, 4, 0, 4, 1, None) == capi.RT_ERR_INVALID
    assert lib.rt_get_timing(None, None) == capi.RT_ERR_INVALID
    assert lib.rt_scene_destroy(None) == capi.RT_OK


def test_more_objects_than_the_index_field_holds_is_a_capacity_error():
    """The item tables carry a 12-bit Scene index: 4 096 objects (the reference's Scene stops at
    3 999, src/Scene.h:8).  More is refused at create time, before any device is touched."""
    host = HostScene.builtin()
    d = _copy_desc(host.desc)
    n = 4097
    objs = (capi.RtObjectDesc * n)(*[d.objects[i % 32] for i in range(n)])
    d.objects, d.n_objects, d.shadow_begin, d.shadow_end = objs, n, 0, n
    rc, msg = _create(d)
    assert rc == capi.RT_ERR_CAPACITY and "4096" in msg


def test_no_gpu_means_no_render(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    from tilecoderaytracer_amd import Renderer
    with pytest.raises(RtError) as e:
        Renderer(HostScene.builtin())
    assert e.value.code == capi.RT_ERR_NO_DEVICE
    lib = capi.load_library()
    host = HostScene.builtin()
    import numpy as np
    out = np.zeros((8, 8, 3), np.float32)
    rc = lib.rt_render_multi(host.desc, host.camera, 8, 8, 1, 1, out.ctypes.data)
    assert rc == capi.RT_ERR_NO_DEVICE and not out.any()


def test_no_gpu_means_no_shared_image(have_gpu):
    """rt_shared_image_* without a device: an error code and a message, no image and no crash; NULL arguments are refused first."""
    if have_gpu:
        pytest.skip("a GPU is present")
    lib = capi.load_library()
    handle = C.create_string_buffer(64)
    p = C.c_void_p(1234)
    assert lib.rt_shared_image_create(0, 1 << 20, None, handle) == capi.RT_ERR_INVALID
    rc = lib.rt_shared_image_create(0, 1 << 20, C.byref(p), handle)
    assert rc != capi.RT_OK and p.value is None and lib.rt_last_error()
    rc = lib.rt_shared_image_open(0, handle.raw, C.byref(p))
    assert rc != capi.RT_OK and p.value is None
    assert lib.rt_shared_image_close(0, None) == capi.RT_OK and lib.rt_shared_image_destroy(0, None) == capi.RT_OK


@pytest.mark.parametrize("name", ["grid32", "grid16", "grid9", "twomirrors"])
def test_packing_a_clustered_scene_needs_no_gpu_and_then_says_so(have_gpu, name):
    """rt_scene_create packs the tables -- sphere clusters, item tables, and for the 1 024-sphere grid the SHADOW VOXELS
    (csrc/rt_capi.hip: shadow_voxels(), 4 096 voxels x 43 leaves x 2 lights) -- on the host before it asks for a device: on a
    machine without a GPU the whole packer runs and the call then fails with RT_ERR_NO_DEVICE, not with a crash."""
    if have_gpu:
        pytest.skip("a GPU is present")
    from tilecoderaytracer_amd import Renderer
    with pytest.raises(RtError) as e:
        Renderer(HostScene.named(name))
    assert e.value.code == capi.RT_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under the package, include/
    or bench.py's product path may import, link or load it."""
    bad = []
    pkg = os.path.join(ROOT, "tilecoderaytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"liboracle|rt_oracle|oracle_lib|orc_render", text):
                    bad.append(os.path.join(dirpath, f))
    # build.py may BUILD the oracle (building the checker is not using it) but not load it
    bad = [b for b in bad if not b.endswith(os.path.join("tilecoderaytracer_amd", "build.py"))]
    assert not bad, bad
    for lib in ("libtcrt.so", "libtcrt_host.so"):
        blob = open(os.path.join(pkg, "lib", lib), "rb").read()
        assert b"orc_render" not in blob and b"liboracle" not in blob


def test_build_keeps_the_arithmetic_contract_flags():
    """Parity depends on how the kernel is compiled: no FMA
    contraction, correctly rounded fp32 divide/sqrt, denormals kept, never fast-math."""
    mk = open(os.path.join(ROOT, "tilecoderaytracer_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^HIPFLAGS\s*:=(.*?)(?:\n\S|\Z)", mk, flags=re.S | re.M).group(1)
    for needed in ("--offload-arch=$(ARCH)", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                   "-fno-gpu-flush-denormals-to-zero", "-fno-fast-math"):
        assert needed in flags, needed
    assert "-ffast-math" not in flags.replace("-fno-fast-math", "")
    assert re.search(r"^ARCH\s*\?=\s*gfx950", mk, flags=re.M)
    kernel = open(os.path.join(ROOT, "tilecoderaytracer_amd", "csrc", "rt_kernel.hip")).read()
    assert "#pragma clang fp contract(off)" in kernel


def test_library_path_can_be_overridden(monkeypatch, tmp_path):
    """TCRT_LIBRARY names another build of libtcrt.so (A/B timing, packaging)."""
    default = capi.library_path()
    assert default.endswith(os.path.join("lib", "libtcrt.so"))
    other = tmp_path / "libtcrt_other.so"
    monkeypatch.setenv("TCRT_LIBRARY", str(other))
    assert capi.library_path() == str(other)
    monkeypatch.delenv("TCRT_LIBRARY")
    assert capi.library_path() == default


def test_counting_build_names_match_the_header():
    from tilecoderaytracer_amd import Renderer
    text = open(os.path.join(ROOT, "include", "rt_capi_tuning.h")).read()
    count = int(re.search(r"#define RT_STATS_COUNT (\d+)", text).group(1))
    assert len(Renderer.STAT_NAMES) == count
