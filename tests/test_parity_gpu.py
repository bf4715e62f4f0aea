"""Parity tests proper: the HIP path, called through the C ABI
(include/rt_capi.h), against the CPU oracle on the same scenes.

Bar: BIT-EXACT (the path is IEEE binary32 with a fixed operation order;
BASELINE.json's 1e-4 per-channel tolerance is the fallback bar and is also
asserted, trivially, by equality).
"""
import numpy as np
import pytest

from tilecoderaytracer_amd import HostScene, Renderer

pytestmark = pytest.mark.gpu


def assert_same(gpu, ref, what):
    assert gpu.shape == ref.shape
    same = gpu.view(np.uint32) == ref.view(np.uint32)
    if not same.all():
        bad = np.argwhere(~same.all(axis=-1))
        diff = np.abs(gpu.astype(np.float64) - ref.astype(np.float64))
        raise AssertionError(
            f"{what}: {len(bad)} pixels differ, max |d|={np.nanmax(diff):.3g}, "
            f"first at {bad[0].tolist()}: gpu={gpu[tuple(bad[0])]} ref={ref[tuple(bad[0])]}")


@pytest.mark.parametrize("name,W,H,depth", [
    ("builtin", 64, 64, 4),
    ("builtin", 256, 256, 4),
    ("builtin", 500, 504, 3),          # the shipped resolution (not a multiple of the tile)
    ("grid16", 128, 128, 8),
    ("grid16-noshadow", 64, 64, 8),
    ("grid32", 96, 96, 4),
    ("grid32-noshadow", 64, 64, 4),
])
def test_scene_bit_exact(oracle, name, W, H, depth):
    ref = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    assert_same(r.render(W, H, depth), ref, f"{name} {W}x{H} d{depth}")


# ---------------------------------------------------------------- fixtures
import hashlib
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("key", ["b64d4", "g32_64d4", "g16_64d8"])
def test_golden_fixtures(oracle, key):
    """Committed fixtures whose SHA-256 SURVEY.md App. D recorded from the reference."""
    name, W, H, depth, digest = oracle.SURVEY_PINS[key]
    want = np.fromfile(os.path.join(GOLDEN, key + ".f32"), dtype=np.float32).reshape(W, H, 3)
    got = Renderer(HostScene.named(name)).render(W, H, depth)
    assert_same(got, want, key)
    assert hashlib.sha256(got.tobytes()).hexdigest() == digest


@pytest.mark.parametrize("key", ["b256d4", "g32d4", "g16d8", "b512d3"])
def test_survey_digests_on_gpu(oracle, key):
    name, W, H, depth, digest = oracle.SURVEY_PINS[key]
    got = Renderer(HostScene.named(name)).render(W, H, depth)
    assert hashlib.sha256(got.tobytes()).hexdigest() == digest


def test_extra_digests():
    extra = json.load(open(os.path.join(GOLDEN, "extra.json")))
    assert "grid32_256x256_d4" in extra and "grid16_256x256_d8" in extra       # the shadowed grids at 256 x 256 (oracle-only pins: "_about")
    for key, digest in extra.items():
        if key.startswith("_"):
            continue
        name, size, d = key.split("_")
        W, H = (int(v) for v in size.split("x"))
        got = Renderer(HostScene.named(name)).render(W, H, int(d[1:]))
        assert hashlib.sha256(got.tobytes()).hexdigest() == digest, key


# ------------------------------------------------------------- edge cases
def test_shipped_configuration_depth_50(oracle):
    """500 x 504, MAX_RECURSION_LEVEL 50 (src/rt_project_parameters.h:65-66,73)."""
    ref = oracle.OracleScene.builtin().render(500, 504, 50)
    assert_same(Renderer(HostScene.builtin()).render(500, 504, 50), ref, "shipped config")


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 5, 8, 13])
def test_depths(oracle, depth):
    ref = oracle.OracleScene.grid(6, True).render(80, 72, depth)
    assert_same(Renderer(HostScene.grid(6, True)).render(80, 72, depth), ref, f"depth {depth}")


def test_two_mirrors_scene_3920_objects(oracle):
    """The reference's own many-sphere scene: 3 920 objects, facing mirrors."""
    ref = oracle.OracleScene.two_mirrors().render(40, 40, 6)
    assert_same(Renderer(HostScene.two_mirrors()).render(40, 40, 6), ref, "two mirrors")


def test_empty_scene_is_background():
    r = Renderer(HostScene.empty())
    img = r.render(33, 17, 4)
    assert (img == np.float32(0.75)).all()


def test_lights_only_and_single_pixel(oracle):
    host, orc = HostScene.empty(), oracle.OracleScene()
    for s in (host, orc):
        i = s.add_sphere((0.0, 4.0, 0.0), 1.5)
        s.set_light(i)
        s.set_intensity(i, 0.5)
    assert_same(Renderer(host).render(1, 1, 4), orc.render(1, 1, 4), "1x1")
    assert_same(Renderer(host).render(31, 3, 4), orc.render(31, 3, 4), "31x3")


def test_empty_strip_and_ragged_sizes(oracle):
    r = Renderer(HostScene.builtin())
    assert r.render(64, 64, 2, 10, 10).shape == (0, 64, 3)
    orc = oracle.OracleScene.builtin()
    for W, H in ((1, 130), (130, 1), (67, 19), (5, 257)):
        assert_same(r.render(W, H, 3), orc.render(W, H, 3), f"{W}x{H}")


def test_strips_are_bit_identical_to_the_full_render(oracle):
    r = Renderer(HostScene.grid(8, True))
    full = r.render(96, 64, 4)
    assert_same(full, oracle.OracleScene.grid(8, True).render(96, 64, 4), "full")
    for x0, x1 in ((0, 24), (24, 48), (48, 96), (95, 96), (13, 14), (1, 95)):
        assert_same(r.render(96, 64, 4, x0, x1), full[x0:x1], f"strip {x0}:{x1}")


@pytest.mark.parametrize("tile_z", [1, 2, 4, 8, 16, 32, 64])
def test_tile_shapes_do_not_change_results(oracle, tile_z):
    r = Renderer(HostScene.builtin())
    r.set_option("tile_z", tile_z)
    assert_same(r.render(150, 70, 4), oracle.OracleScene.builtin().render(150, 70, 4), f"tile_z {tile_z}")


@pytest.mark.parametrize("first_row", [-1, 0, 333, 500, 999])
def test_tile_queue_start_row_does_not_change_results(oracle, first_row):
    """Where the tile queues start (rt_set_option "first_row"; automatic = just below the
    horizon row for scenes with clustered sphere runs) is scheduling only."""
    from tilecoderaytracer_amd import RtError
    r = Renderer(HostScene.grid(9, True))                 # 81 consecutive spheres: a clustered run
    r.set_option("first_row", first_row)
    want = oracle.OracleScene.grid(9, True).render(120, 200, 3)
    assert_same(r.render(120, 200, 3), want, f"first_row {first_row}")
    assert_same(r.render(120, 200, 3, 17, 60), want[17:60], f"first_row {first_row}, strip")
    with pytest.raises(RtError):
        r.set_option("first_row", 1000)


@pytest.mark.parametrize("block", [64, 128, 192, 256])
def test_block_sizes_do_not_change_results(oracle, block):
    r = Renderer(HostScene.grid(5, True))
    r.set_option("block_threads", block)
    assert_same(r.render(90, 50, 6), oracle.OracleScene.grid(5, True).render(90, 50, 6), f"block {block}")
    assert r.launch_info().block_threads == block


@pytest.mark.parametrize("block", [320, 384, 512])
def test_wide_workgroups_of_the_clustered_scene_kernel(oracle, block):
    """The clustered-scene kernel takes workgroups of up to eight wavefronts (one LDS copy of the tables for twice as
    many; launch() picks 512 threads by itself for scenes whose tables are large).  Frames and strips, HELP from two
    leaves on and the HEAVY band on, so that up to seven wavefronts stand at one desk."""
    want = oracle.OracleScene.named("grid16").render(96, 160, 8)
    r = Renderer(HostScene.named("grid16"))
    r.set_option("block_threads", block)
    assert_same(r.render(96, 160, 8), want, f"grid16 block {block}")
    assert r.launch_info().block_threads == block and r.launch_info().kernel.startswith(b"rt_render_kernel_clusters")
    r.set_option("help", 2)
    r.set_option("heavy", 3)
    for _ in range(2):
        assert_same(r.render(96, 160, 8), want, f"grid16 block {block}, help and heavy")
    assert_same(r.render(96, 160, 8, 20, 77), want[20:77], f"grid16 block {block}, help and heavy, strip")


def test_large_tables_get_workgroups_of_eight_wavefronts(oracle):
    """The 1 024-sphere grid: 28 KB of tables.  Four-wavefront workgroups fit five times per CU (five wavefronts per SIMD, the
    96-register kernel); launch() takes eight-wavefront workgroups instead, three per CU = six wavefronts per SIMD in the
    80-register kernel, with bounce-stack levels in LDS.  Same pixels; an explicit block_threads still wins."""
    want = oracle.OracleScene.named("grid32").render(160, 96, 4)
    r = Renderer(HostScene.named("grid32"))
    assert_same(r.render(160, 96, 4), want, "grid32, automatic workgroup")
    li = r.launch_info()
    assert li.block_threads == 512 and li.kernel == b"rt_render_kernel_clusters" and li.lds_bytes * 3 <= 160 * 1024
    assert li.lds_bytes > li.scene_lds_bytes                      # stack levels in LDS
    assert_same(r.render(160, 96, 4, 16, 48), want[16:48], "grid32, automatic workgroup, strip (HELP, HEAVY)")
    r.set_option("block_threads", 256)
    assert_same(r.render(160, 96, 4), want, "grid32, 256 threads")
    assert r.launch_info().block_threads == 256 and r.launch_info().kernel == b"rt_render_kernel_clusters_wide"
    r16 = Renderer(HostScene.named("grid16"))                      # small tables: four-wavefront workgroups, as before
    r16.render(64, 64, 2)
    assert r16.launch_info().block_threads == 256 and r16.launch_info().kernel == b"rt_render_kernel_clusters"


@pytest.mark.parametrize("seed", range(1, 13))
def test_random_scenes(oracle, seed):
    """Mixed primitives in shuffled index order, textured finite planes, random
    materials -- exercises run splitting, tie order and every shading branch."""
    from scene_gen import build_random
    host = build_random(HostScene.empty(), seed, shadows=(seed % 3 != 0))
    orc = build_random(oracle.OracleScene(), seed, shadows=(seed % 3 != 0))
    assert_same(Renderer(host).render(72, 56, 5), orc.render(72, 56, 5), f"seed {seed}")


@pytest.mark.parametrize("seed", range(200, 224))
def test_axis_aligned_rooms(oracle, seed):
    """Boxes of axis-aligned rectangles, axis-aligned infinite planes (slabs), lights hugging walls, scales from 1e-2 to
    1e3: the tight plane boxes and slabs of the culls against the oracle, FAST tables and item tables."""
    from scene_gen import build_room
    host, orc = build_room(HostScene.empty(), seed), build_room(oracle.OracleScene(), seed)
    want = orc.render(88, 72, 5)
    r = Renderer(host)
    assert_same(r.render(88, 72, 5), want, f"room seed {seed}")
    if seed % 3 == 0:
        r.set_option("fast", 0)
        assert_same(r.render(88, 72, 5), want, f"room seed {seed}, item tables")


@pytest.mark.parametrize("tile_z", [0, 1, 4, 64])
def test_primary_table_against_the_bundle_cull(oracle, tile_z):
    """FAST tables: the camera rays' scan culls by the items' projected pixel rectangles (option primary = 1, the default) or
    by the bundle like every other scan (0): same image, for whole frames and strips, the reference's two cameras, a rolled one,
    and an eye that sits inside a sphere (that item gets the whole image)."""
    from scene_gen import build_room
    cases = [(HostScene.builtin(), oracle.OracleScene.builtin(), 150, 120, 4)]
    cases.append((build_room(HostScene.empty(), 206), build_room(oracle.OracleScene(), 206), 97, 61, 5))
    host, orc = build_room(HostScene.empty(), 210), build_room(oracle.OracleScene(), 210)
    for sc in (host, orc):
        i = sc.add_sphere((0.0, -1.0, 2.5), 3.0)          # around the eye of the two-mirrors camera
        sc.set_reflective(i, 0.5)
    cases.append((host, orc, 64, 80, 4))
    host, orc = build_room(HostScene.empty(), 211), build_room(oracle.OracleScene(), 211)
    hc = host.camera.contents
    h = np.array(list(hc.vector_horizontal), dtype=np.float32); v = np.array(list(hc.vector_vertical), dtype=np.float32)
    ca, sa = np.float32(np.cos(0.4)), np.float32(np.sin(0.4))
    h2, v2 = ca * h + sa * v, ca * v - sa * h
    for k in range(3):
        hc.vector_horizontal[k], hc.vector_vertical[k] = float(h2[k]), float(v2[k])
    orc.cam.vector_horizontal = type(orc.cam.vector_horizontal)(*[float(c) for c in h2])
    orc.cam.vector_vertical = type(orc.cam.vector_vertical)(*[float(c) for c in v2])
    cases.append((host, orc, 90, 70, 4))
    for host, orc, W, H, depth in cases:
        want = orc.render(W, H, depth)
        r = Renderer(host)
        if tile_z:
            r.set_option("tile_z", tile_z)
        for primary in (1, 0):
            r.set_option("primary", primary)
            assert_same(r.render(W, H, depth), want, f"primary {primary}, tile_z {tile_z}")
            assert_same(r.render(W, H, depth, 13, W - 7), want[13:W - 7], f"primary {primary}, tile_z {tile_z}, strip")


def test_partial_shadow_range(oracle):
    """Scene::SetObjectIndices(rank, size) narrows the shadow scan (src/Scene.cpp:486-504)."""
    from scene_gen import build_random
    for rank, size in ((0, 2), (1, 2), (2, 3)):
        host = build_random(HostScene.empty(), 5, shadows=False)
        orc = build_random(oracle.OracleScene(), 5, shadows=False)
        host.set_object_indices(rank, size)
        orc.set_object_indices(rank, size)
        assert_same(Renderer(host).render(64, 48, 4), orc.render(64, 48, 4), f"indices {rank}/{size}")


def test_absurd_depth_is_an_error_not_a_fallback():
    from tilecoderaytracer_amd import RtError, capi
    r = Renderer(HostScene.builtin())
    with pytest.raises(RtError) as e:
        r.render(8, 8, 2000000000)          # the bounce stack would need > 8 GB of HBM
    assert e.value.code == capi.RT_ERR_CAPACITY


def test_facing_mirrors_depth_400(oracle):
    """Two perfect mirrors facing each other: every level of the bounce stack is used."""
    host, orc = HostScene.empty(), oracle.OracleScene()
    for s in (host, orc):
        i = s.add_sphere((3.0, 5.0, 8.0), 0.15)
        s.set_light(i)
        a = s.add_finite_plane_axes((-4.0, 9.0, -1.0), (0.0, -1.0, 0.0), (1.0, 0.0, 0.0), 8.0, 8.0)
        b = s.add_finite_plane_axes((4.0, -3.0, -1.0), (0.0, 1.0, 0.0), (-1.0, 0.0, 0.0), 8.0, 8.0)
        for m in (a, b):
            s.set_reflective(m, 1.0)
            s.set_diffuse(m, 0.0)
        k = s.add_sphere((0.5, 3.0, 2.0), 0.7)
        s.set_color(k, (1, 0, 0))
        s.set_object_indices(0, 1)
        s.camera_two_mirrors()
    assert_same(Renderer(host).render(48, 40, 400), orc.render(48, 40, 400), "facing mirrors d400")


@pytest.mark.parametrize("width,height", [(3.0, 3.0), (0.3, 7.7), (1.0e-3, 2.5), (1.0e5, 0.75), (2.0 ** -110, 1.0),
                                          (1.0, 2.0 ** 101), (0.0, 1.0), (-2.0, 3.0)])
def test_checkerboard_coordinates(oracle, width, height):
    """Texture_CheckerBoard::getTexturePixel (src/Texture_CheckerBoard.h:31-65) on an
    infinite plane seen to the horizon: negative and positive coordinates, quotients
    from < 1 to > 2^20, degenerate cell sizes -- the kernel's short fmodf route and the
    library route it falls back to must both give the reference's cells."""
    host, orc = HostScene.empty(), oracle.OracleScene()
    for s in (host, orc):
        i = s.add_sphere((2.0, -3.0, 9.0), 0.15)
        s.set_light(i)
        g = s.add_infinite_plane((0.3, 0.1, -1.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
        s.set_checkerboard(g, (1.0, 1.0, 1.0), (0.0, 0.0, 0.25), width, height)
        s.set_reflective(g, 0.25)
        k = s.add_sphere((0.5, 6.0, 0.5), 1.5)
        s.set_reflective(k, 1.0)
        s.set_object_indices(0, 1)
    assert_same(Renderer(host).render(96, 64, 3), orc.render(96, 64, 3), f"checkerboard {width} x {height}")


# ------------------------------------------- full-size (BASELINE.json sizes)
def test_4096_builtin_depth4_properties(oracle):
    """configs[1] at full size.  (a) (float)(64k)/4096 == (float)k/64, so the
    stride-64 subsample must equal the committed 64x64 fixture; (b) columns at
    odd offsets are compared with the oracle directly; (c) determinism."""
    r = Renderer(HostScene.builtin())
    img = r.render(4096, 4096, 4)
    want = np.fromfile(os.path.join(GOLDEN, "b64d4.f32"), dtype=np.float32).reshape(64, 64, 3)
    assert_same(np.ascontiguousarray(img[::64, ::64]), want, "stride-64 subsample")
    orc = oracle.OracleScene.builtin()
    for x0 in (1, 1001, 2047, 3333, 4095):
        assert_same(img[x0:x0 + 1], orc.render(4096, 4096, 4, x0, x0 + 1), f"column {x0}")
    assert hashlib.sha256(img.tobytes()).hexdigest() == hashlib.sha256(r.render(4096, 4096, 4).tobytes()).hexdigest()
    assert np.isfinite(img).all() and img.max() > 1.0


def _oracle_columns(oracle, make_scene, W, H, depth, columns, threads=8):
    """{x: oracle column x} rendered by a few threads (the C oracle releases the GIL; one scene per thread)."""
    from concurrent.futures import ThreadPoolExecutor
    columns = list(columns)
    chunks = [columns[i::threads] for i in range(threads) if columns[i::threads]]

    def work(xs):
        scene = make_scene()
        return {x: scene.render(W, H, depth, x, x + 1) for x in xs}

    out = {}
    with ThreadPoolExecutor(max_workers=len(chunks)) as pool:
        for part in pool.map(work, chunks):
            out.update(part)
    return out


def test_4096_builtin_depth4_every_pixel(oracle):
    """configs[1], the bench workload, compared with the oracle pixel for pixel (16.7 M pixels)."""
    img = Renderer(HostScene.builtin()).render(4096, 4096, 4)
    from concurrent.futures import ThreadPoolExecutor
    bounds = [(k * 256, (k + 1) * 256) for k in range(16)]

    def work(b):
        return oracle.OracleScene.builtin().render(4096, 4096, 4, b[0], b[1])

    with ThreadPoolExecutor(max_workers=8) as pool:
        for (x0, x1), want in zip(bounds, pool.map(work, bounds)):
            assert_same(img[x0:x1], want, f"columns {x0}:{x1}")


def test_4096_grid32_depth4_columns(oracle):
    """configs[2] at full size: sampled columns against the oracle + the 64x64 subsample."""
    r = Renderer(HostScene.grid(32, True))
    img = r.render(4096, 4096, 4)
    orc = oracle.OracleScene.grid(32, True)
    assert_same(np.ascontiguousarray(img[::64, ::64]), orc.render(64, 64, 4), "stride-64 subsample")
    # whole columns: every row of them, the horizon rows (far hit points) included
    cols = sorted(set([777, 2049] + list(range(5, 4096, 131))))
    for x0, want in _oracle_columns(oracle, lambda: oracle.OracleScene.grid(32, True), 4096, 4096, 4, cols).items():
        assert_same(img[x0:x0 + 1], want, f"column {x0}")


def test_4096_grid16_depth8_columns(oracle):
    """configs[4] at full size."""
    r = Renderer(HostScene.grid(16, True))
    img = r.render(4096, 4096, 8)
    orc = oracle.OracleScene.grid(16, True)
    assert_same(np.ascontiguousarray(img[::64, ::64]), orc.render(64, 64, 8), "stride-64 subsample")
    cols = sorted(set([123, 3001] + list(range(17, 4096, 257))))
    for x0, want in _oracle_columns(oracle, lambda: oracle.OracleScene.grid(16, True), 4096, 4096, 8, cols).items():
        assert_same(img[x0:x0 + 1], want, f"column {x0}")


def _assert_every_pixel(oracle, img, make_oracle_scene, depth, what, size=4096):
    """All size x size pixels of `img` against the oracle, 64-column blocks dealt to the host's cores
    (the C oracle releases the GIL; one scene per worker call)."""
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    bounds = [(k * 64, (k + 1) * 64) for k in range(size // 64)]

    def work(b):
        return make_oracle_scene().render(size, size, depth, b[0], b[1])

    with ThreadPoolExecutor(max_workers=workers) as pool:
        for (x0, x1), want in zip(bounds, pool.map(work, bounds)):
            assert_same(img[x0:x1], want, f"{what}: columns {x0}:{x1}")


def test_4096_grid32_depth4_every_pixel(oracle):
    """configs[2] (1 024 spheres, shadow scan on), the frame bench.py's `sphere_grid` line times: all 16.7 M pixels
    against the oracle (about 10 core-minutes of oracle: 40 s on the GPU box's 16 cores)."""
    img = Renderer(HostScene.grid(32, True)).render(4096, 4096, 4)
    _assert_every_pixel(oracle, img, lambda: oracle.OracleScene.grid(32, True), 4, "grid-32 depth 4")


def test_4096_grid16_depth8_every_pixel(oracle):
    """configs[4] (256 spheres, depth 8): all 16.7 M pixels against the oracle."""
    img = Renderer(HostScene.grid(16, True)).render(4096, 4096, 8)
    _assert_every_pixel(oracle, img, lambda: oracle.OracleScene.grid(16, True), 8, "grid-16 depth 8")


def test_4096_two_mirrors_depth4_every_pixel(oracle):
    """The reference's SCENE 2 (3 920 objects; `bench.py --workload twomirrors`, the global-memory tables' kernel) at 4096^2, depth 4:
    all 16.7 M pixels against the oracle (about a minute of the box's 16 cores)."""
    r = Renderer(HostScene.named("twomirrors"))
    img = r.render(4096, 4096, 4)
    assert r.launch_info().kernel == b"rt_render_kernel_large"
    _assert_every_pixel(oracle, img, lambda: oracle.OracleScene.named("twomirrors"), 4, "two mirrors depth 4")


def test_render_device_into_torch_memory(oracle):
    """The device-pointer entry point used by bench.py and the multi-GPU path."""
    import torch
    host = HostScene.builtin()
    r = Renderer(host)
    buf = torch.zeros((40, 48, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    r.render_device(80, 48, 4, 20, 60, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    ref = oracle.OracleScene.builtin().render(80, 48, 4, 20, 60)
    assert_same(buf.cpu().numpy(), ref, "render_device")
    assert r.timing().launches >= 1


def test_render_multi_single_gpu(oracle):
    """rt_render_multi with ngpu = 1 (the only size a one-GPU box can run)."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    host = HostScene.builtin()
    out = np.zeros((50, 30, 3), np.float32)
    capi.check(capi.load_library().rt_render_multi(host.desc, host.camera, 50, 30, 3, 1, out.ctypes.data))
    assert_same(out, oracle.OracleScene.builtin().render(50, 30, 3), "rt_render_multi")
    rc = capi.load_library().rt_render_multi(host.desc, host.camera, 50, 30, 3, 64, out.ctypes.data)
    assert rc == capi.RT_ERR_INVALID


@pytest.mark.parametrize("seed", range(300, 312))
def test_render_multi_random_scenes_and_shapes_on_one_device(oracle, monkeypatch, seed):
    """rt_render_multi (create + measure + cut + render + destroy) with 2 ... 7 strips on the box's one device
    (TCRT_MULTI_ONE_DEVICE=1), random scenes, image shapes down to fewer columns than GPUs: the strip arithmetic, the measured
    cut and the direct stores on ragged sizes."""
    import ctypes as C
    from scene_gen import build_random, build_sphere_field
    from tilecoderaytracer_amd import capi
    monkeypatch.setenv("TCRT_MULTI_ONE_DEVICE", "1")
    rng = np.random.RandomState(seed)
    ngpu = int(rng.choice([2, 3, 4, 5, 7]))
    W = int(rng.choice([1, 3, ngpu - 1, ngpu, 17, 64, 150]))
    H, depth = int(rng.randint(1, 70)), int(rng.randint(0, 6))
    if seed % 3 == 0:
        mk = lambda s: build_sphere_field(s, seed, n_spheres=130, spread=60.0)
    else:
        mk = lambda s: build_random(s, seed, shadows=(seed % 2 == 0))
    host, orc = mk(HostScene.empty()), mk(oracle.OracleScene())
    out = np.full((W, H, 3), -3.0, np.float32)
    capi.check(capi.load_library().rt_render_multi(host.desc, host.camera, W, H, depth, ngpu, out.ctypes.data))
    assert_same(out, orc.render(W, H, depth), f"seed {seed}: {ngpu} strips of a {W} x {H} image, depth {depth}")


@pytest.mark.parametrize("ngpu", [2, 3, 5])
def test_multi_handle_strip_buffer_transport_and_the_trial_on_one_device(oracle, monkeypatch, ngpu):
    """The strip-buffer transport of the one-process path with SEVERAL strips -- per-GPU strip buffers, column chunks, a kernel-done
    event per chunk, the transfer on the GPU's communication stream while the next chunk renders, the measured cut with its send
    term -- and the automatic choice between it and the direct stores (two frames of each on its own cut), all on the box's one
    device: TCRT_MULTI_ONE_DEVICE=2 stands local device-to-device copies in for ncclSend / ncclRecv (csrc/rt_multi.hip, LOOPBACK).
    RCCL itself needs two GPUs."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    monkeypatch.setenv("TCRT_MULTI_ONE_DEVICE", "2")
    host, orc = HostScene.named("grid16"), oracle.OracleScene.named("grid16")
    W, H, depth = 333, 80, 5
    want = orc.render(W, H, depth)
    m = C.c_void_p()
    capi.check(lib.rt_multi_create(host.desc, ngpu, C.byref(m)))
    try:
        info = capi.RtMultiInfo()
        out = np.zeros((W, H, 3), dtype=np.float32)
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))            # automatic: both measured, the trial
        assert_same(out, want, "automatic transport")
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert info.transport in (1, 2) and info.balanced == 1 and info.trial_frame_ms[0] > 0.0 and info.trial_frame_ms[1] > 0.0
        assert (info.transport == 2) == (info.trial_frame_ms[1] < info.trial_frame_ms[0])
        assert info.trial_image_ok[0] == 1 and info.trial_image_ok[1] == 1        # both images equalled GPU 0's own on the sampled columns
        assert info.bounds[0] == 0 and info.bounds[ngpu] == W
        for transport in (1, 2, 1):
            capi.check(lib.rt_multi_set_option(m, b"transport", transport))
            out[:] = 0
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))        # re-measured for that transport
            assert_same(out, want, f"transport {transport}, measured cut")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.transport == transport and info.balanced == 1 and 1 <= info.chunks <= 8
            if transport == 1:
                assert info.measured_gather_ms > 0.0
        # strip buffers with an explicit chunk count on equal strips, then the caller's own cut (an empty strip in it) in 7 chunks
        for chunks in (1, 5, 64):
            out[:] = 0
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, chunks, out.ctypes.data))
            assert_same(out, want, f"equal strips, {chunks} chunks")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.transport == 1 and info.chunks == chunks and info.balanced == 0
        cut = [0] + [41] * (ngpu - 1) + [W]
        capi.check(lib.rt_multi_set_bounds(m, W, (C.c_int * (ngpu + 1))(*cut), 7))
        out[:] = 0
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, want, "caller's strips in 7 chunks")
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert [info.bounds[g] for g in range(ngpu + 1)] == cut and info.chunks == 7 and info.transport == 1
    finally:
        capi.check(lib.rt_multi_destroy(m))


def test_threads_share_a_handle_and_use_their_own(oracle):
    """calls on ONE handle from several threads serialise on its mutex (every thread gets its own
    strip, whole and exact); handles of different threads are independent (ctypes releases the GIL: the calls really overlap)."""
    from concurrent.futures import ThreadPoolExecutor
    W, H, depth = 160, 120, 5
    shared = Renderer(HostScene.builtin())
    want_builtin = oracle.OracleScene.builtin().render(W, H, depth)

    def strip_on_the_shared_handle(k):
        x0, x1 = k * 20, (k + 1) * 20
        for _ in range(6):
            got = shared.render(W, H, depth, x0, x1)
            if not np.array_equal(got.view(np.uint32), want_builtin[x0:x1].view(np.uint32)):
                return f"strip {k} of the shared handle"
        return None

    names = ["builtin", "grid9", "grid16", "twomirrors"]
    wants = {n: oracle.OracleScene.named(n).render(96, 64, 4) for n in names}

    def own_handle(k):
        name = names[k % len(names)]
        r = Renderer(HostScene.named(name))
        for _ in range(4):
            got = r.render(96, 64, 4)
            if not np.array_equal(got.view(np.uint32), wants[name].view(np.uint32)):
                return f"{name} on thread {k}'s own handle"
        return None

    with ThreadPoolExecutor(max_workers=16) as pool:
        jobs = [pool.submit(strip_on_the_shared_handle, k) for k in range(8)] + [pool.submit(own_handle, k) for k in range(8)]
        wrong = [j.result() for j in jobs if j.result()]
    assert not wrong, wrong


@pytest.mark.parametrize("spoiled", ["rccl", "direct", "both"])
def test_multi_handle_does_not_choose_a_transport_whose_image_is_wrong(oracle, monkeypatch, spoiled):
    """Correctness before speed in rt_multi_render's automatic choice: with one pixel of a transport's trial image spoiled
    (TCRT_MULTI_CORRUPT, a testing aid) the OTHER transport is used whatever the times were; with both spoiled the call fails
    loudly instead of delivering either."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    monkeypatch.setenv("TCRT_MULTI_ONE_DEVICE", "2")
    host, orc = HostScene.named("grid9"), oracle.OracleScene.named("grid9")
    W, H, depth = 200, 64, 3
    m = C.c_void_p()
    capi.check(lib.rt_multi_create(host.desc, 3, C.byref(m)))
    try:
        out = np.zeros((W, H, 3), dtype=np.float32)
        info = capi.RtMultiInfo()
        if spoiled == "both":
            monkeypatch.setenv("TCRT_MULTI_CORRUPT", "both")
            assert lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data) == capi.RT_ERR_HIP
            assert b"neither transport" in lib.rt_last_error() and not out.any()
            monkeypatch.delenv("TCRT_MULTI_CORRUPT")                   # the handle is usable afterwards: it measures again
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
            assert_same(out, orc.render(W, H, depth), "the frame after a failed choice")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.trial_image_ok[0] == 1 and info.trial_image_ok[1] == 1
            return
        monkeypatch.setenv("TCRT_MULTI_CORRUPT", spoiled)
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, orc.render(W, H, depth), "the frame after the choice")
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert info.transport == (2 if spoiled == "rccl" else 1)
        assert (info.trial_image_ok[0], info.trial_image_ok[1]) == ((0, 1) if spoiled == "rccl" else (1, 0))
    finally:
        capi.check(lib.rt_multi_destroy(m))


def test_render_multi_the_references_simulator_configuration(oracle, monkeypatch):
    """IS_FOR_SIMULATION (src/rt_project_parameters.h:45-52, src/RayTracer.h:64-66): 2 cores, 5 x 5 pixels -- the reference's own
    way of running its parallel path without the hardware; here two strips on the one device, the shipped depth 50."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    monkeypatch.setenv("TCRT_MULTI_ONE_DEVICE", "1")
    host = HostScene.builtin()
    out = np.zeros((5, 5, 3), np.float32)
    capi.check(capi.load_library().rt_render_multi(host.desc, host.camera, 5, 5, 50, 2, out.ctypes.data))
    assert_same(out, oracle.OracleScene.builtin().render(5, 5, 50), "2 cores, 5 x 5 pixels")


def test_multi_handle_renders_frames_in_column_chunks(oracle):
    """rt_multi_create / rt_multi_render / rt_multi_destroy: scenes, streams, buffers (and, beyond one GPU, the
    communicator) persist across frames; every frame is rendered in `chunks` launches whose column chunks land in
    place.  One GPU is all this box has: the transfers are then nothing, the chunked renders and the handle are
    what is exercised -- sizes, depths and chunk counts change from frame to frame on the same handle."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    for name in ("builtin", "grid9"):
        host, orc = HostScene.named(name), oracle.OracleScene.named(name)
        m = C.c_void_p()
        capi.check(lib.rt_multi_create(host.desc, 1, C.byref(m)))
        try:
            capi.check(lib.rt_multi_set_option(m, b"help", 2))
            for W, H, depth, chunks in ((50, 30, 3, 1), (200, 64, 4, 4), (37, 19, 2, 8), (200, 64, 4, 64), (16, 16, 1, 3)):
                out = np.zeros((W, H, 3), dtype=np.float32)
                capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, chunks, out.ctypes.data))
                assert_same(out, orc.render(W, H, depth), f"{name} {W}x{H} d{depth} in {chunks} chunks")
            out = np.zeros((8, 8, 3), dtype=np.float32)
            assert lib.rt_multi_render(m, host.camera, 8, 8, 1, -1, out.ctypes.data) == capi.RT_ERR_INVALID
            assert lib.rt_multi_render(m, host.camera, 8, 8, 1, 65, out.ctypes.data) == capi.RT_ERR_INVALID
        finally:
            capi.check(lib.rt_multi_destroy(m))
    assert lib.rt_multi_create(host.desc, 64, C.byref(m)) == capi.RT_ERR_INVALID      # more GPUs than the box has


def test_multi_handle_automatic_partition_and_callers_bounds(oracle):
    """rt_multi_render with chunks = 0 (what rt_render_multi and tcrt_raytracer --gpus use): the strips come from a
    measurement of the frame shape, made once and reused, or from rt_multi_set_bounds.  On one GPU the cut is the whole
    image (nothing to balance, nothing to send); what is checked is the path, the bookkeeping rt_multi_get_info reports,
    and that a frame that FAILS half-way leaves the handle usable (nothing queued, no RCCL group open)."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    host, orc = HostScene.named("grid9"), oracle.OracleScene.named("grid9")
    m = C.c_void_p()
    capi.check(lib.rt_multi_create(host.desc, 1, C.byref(m)))
    try:
        info = capi.RtMultiInfo()
        for W, H, depth in ((96, 40, 3), (96, 40, 3), (50, 30, 2)):                 # the second frame reuses the first one's cut
            out = np.zeros((W, H, 3), dtype=np.float32)
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
            assert_same(out, orc.render(W, H, depth), f"automatic partition {W}x{H}")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.ngpu == 1 and info.chunks == 1 and info.balanced == 0
            assert (info.bounds[0], info.bounds[1]) == (0, W) and info.kernel_ms[0] > 0.0 and info.frame_ms >= info.kernel_ms[0]
        # a caller's own cut: one strip [0, W) in 5 chunks; a cut for another width is ignored for this one
        W, H, depth = 96, 40, 3
        bounds = (C.c_int * 2)(0, W)
        capi.check(lib.rt_multi_set_bounds(m, W, bounds, 5))
        out = np.zeros((W, H, 3), dtype=np.float32)
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, orc.render(W, H, depth), "caller's bounds")
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert info.chunks == 5
        bad = (C.c_int * 2)(0, W - 1)
        assert lib.rt_multi_set_bounds(m, W, bad, 1) == capi.RT_ERR_INVALID
        bad = (C.c_int * 2)(3, W)
        assert lib.rt_multi_set_bounds(m, W, bad, 1) == capi.RT_ERR_INVALID
        assert lib.rt_multi_set_bounds(m, W, bounds, 0) == capi.RT_ERR_INVALID
        capi.check(lib.rt_multi_set_bounds(m, 0, None, 0))                            # forget it
        # a frame that fails after the handle has queued work before (a negative depth is refused by the first launch)
        assert lib.rt_multi_render(m, host.camera, W, H, -1, 3, out.ctypes.data) == capi.RT_ERR_INVALID
        assert b"max_depth" in lib.rt_last_error()
        assert lib.rt_multi_render(m, host.camera, W, H, -1, 0, out.ctypes.data) == capi.RT_ERR_INVALID
        out[:] = 0
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, orc.render(W, H, depth), "after two failed frames")
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 4, out.ctypes.data))
        assert_same(out, orc.render(W, H, depth), "after two failed frames, chunked")
        # the handle's own option: how the strips reach device 0 (with one GPU there is nothing to carry: the setting is
        # accepted, re-measures, and changes no pixel; rt_multi_info says RCCL, the transport of a handle without peers)
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert info.transport == 1 and info.trial_frame_ms[0] == 0.0 and info.trial_frame_ms[1] == 0.0
        for transport in (2, 1, 0):
            capi.check(lib.rt_multi_set_option(m, b"transport", transport))
            out[:] = 0
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
            assert_same(out, orc.render(W, H, depth), f"transport {transport}")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.transport == 1 and info.chunks == 1
        assert lib.rt_multi_set_option(m, b"transport", 3) == capi.RT_ERR_INVALID and b"transport" in lib.rt_last_error()
        assert lib.rt_multi_set_option(m, b"transport", -1) == capi.RT_ERR_INVALID
    finally:
        capi.check(lib.rt_multi_destroy(m))


@pytest.mark.parametrize("ngpu", [2, 3])
def test_multi_handle_direct_stores_several_strips_on_one_device(oracle, monkeypatch, ngpu):
    """The DIRECT transport of the one-process multi-GPU path -- every GPU's kernel stores its strip straight into the image on
    device 0 -- with ngpu strips, scenes, streams and the measured cut, all on the box's one device (TCRT_MULTI_ONE_DEVICE=1, a
    testing aid read by rt_multi_create: no RCCL, which refuses two ranks on a device).  Peer access itself needs two GPUs."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    monkeypatch.setenv("TCRT_MULTI_ONE_DEVICE", "1")
    host, orc = HostScene.named("grid16"), oracle.OracleScene.named("grid16")
    W, H, depth = 200, 96, 6
    want = orc.render(W, H, depth)
    m = C.c_void_p()
    capi.check(lib.rt_multi_create(host.desc, ngpu, C.byref(m)))
    try:
        info = capi.RtMultiInfo()
        for _ in range(2):                                    # the second frame reuses the measured cut
            out = np.zeros((W, H, 3), dtype=np.float32)
            capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
            assert_same(out, want, f"{ngpu} strips, direct stores, measured cut")
            capi.check(lib.rt_multi_get_info(m, C.byref(info)))
            assert info.ngpu == ngpu and info.transport == 2 and info.balanced == 1 and info.chunks == 1
            assert info.bounds[0] == 0 and info.bounds[ngpu] == W and all(info.bounds[g] <= info.bounds[g + 1] for g in range(ngpu))
            assert all(info.measured_kernel_ms[g] > 0.0 for g in range(ngpu)) and info.measured_gather_ms == 0.0
            assert info.trial_frame_ms[0] == 0.0 and info.trial_frame_ms[1] == 0.0      # nothing to choose between: no RCCL here
            assert all(info.kernel_ms[g] > 0.0 for g in range(ngpu) if info.bounds[g + 1] > info.bounds[g])
        # the caller's own strips (one of them empty, none on a tile boundary) and an explicit chunk count: still direct stores
        cut = [0, 37, 37, W] if ngpu == 3 else [0, 123, W]
        capi.check(lib.rt_multi_set_bounds(m, W, (C.c_int * (ngpu + 1))(*cut), 4))
        out = np.zeros((W, H, 3), dtype=np.float32)
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, want, "caller's strips")
        capi.check(lib.rt_multi_get_info(m, C.byref(info)))
        assert [info.bounds[g] for g in range(ngpu + 1)] == cut and info.transport == 2 and info.chunks == 1
        capi.check(lib.rt_multi_set_bounds(m, 0, None, 0))
        out[:] = 0
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 3, out.ctypes.data))       # equal strips
        assert_same(out, want, "equal strips")
        assert lib.rt_multi_set_option(m, b"transport", 1) == capi.RT_ERR_INVALID and b"RCCL" in lib.rt_last_error()
        capi.check(lib.rt_multi_set_option(m, b"transport", 2))
        out[:] = 0
        capi.check(lib.rt_multi_render(m, host.camera, W, H, depth, 0, out.ctypes.data))
        assert_same(out, want, "transport 2")
    finally:
        capi.check(lib.rt_multi_destroy(m))


def test_host_executable_writes_the_reference_log(oracle, tmp_path):
    """tcrt_raytracer = the reference's main(): default run writes raytracer_screen.txt."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "tilecoderaytracer_amd", "bin", "tcrt_raytracer")
    out = tmp_path / "raytracer_screen.txt"
    subprocess.run([exe, "--width", "64", "--height", "48", "--depth", "3", "--out", str(out)], check=True,
                   stdout=subprocess.PIPE, cwd=tmp_path)
    ref = oracle.OracleScene.builtin().render(64, 48, 3)
    want = tmp_path / "want.txt"
    oracle.write_screen_txt(str(want), ref, 0.0, 0.0)
    got_lines = out.read_bytes().split(b"\n")
    want_lines = want.read_bytes().split(b"\n")
    assert got_lines[:7] == want_lines[:7] and got_lines[9:] == want_lines[9:]   # only the two timing lines vary


def test_host_executable_with_three_strips_on_one_device(oracle, tmp_path):
    """tcrt_raytracer --gpus 3 (the reference's CORE_NUM) with TCRT_MULTI_ONE_DEVICE=1: three strips cut by measured cost, stored
    by their kernels straight into GPU 0's image; the pixel lines are those of the one-GPU run, the log names the partition."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "tilecoderaytracer_amd", "bin", "tcrt_raytracer")
    out = tmp_path / "raytracer_screen.txt"
    run = subprocess.run([exe, "--width", "150", "--height", "64", "--depth", "5", "--gpus", "3", "--out", str(out)], check=True,
                         stdout=subprocess.PIPE, cwd=tmp_path, env=dict(os.environ, TCRT_MULTI_ONE_DEVICE="1"))
    text = run.stdout.decode()
    assert "Partition: 3 x-strips of" in text and "cut by measured cost" in text and "straight into GPU 0's image" in text
    want = tmp_path / "want.txt"
    oracle.write_screen_txt(str(want), oracle.OracleScene.builtin().render(150, 64, 5), 0.0, 0.0)
    got_lines, want_lines = out.read_bytes().split(b"\n"), want.read_bytes().split(b"\n")
    got_pixels, want_pixels = [l for l in got_lines if l[:1] == b"("], [l for l in want_lines if l[:1] == b"("]
    assert len(want_pixels) == 150 * 64 and got_pixels == want_pixels
    assert b"Number_of_Cores:3." in got_lines


# ------------------------------------------------ stress for the exact culls
def _adversarial(scene, seed):
    """Geometry chosen to sit on the edges of the conservative culls: tiny and
    huge spheres, spheres far from the camera, skewed (non-orthogonal) finite
    planes, boxes (axis-aligned rectangles), coincident surfaces (distance
    ties), lights inside clusters, a partial shadow range."""
    rng = np.random.RandomState(seed)
    f = lambda x: float(np.float32(x))
    lights = [((f(rng.uniform(-30, 30)), f(rng.uniform(0, 60)), f(rng.uniform(3, 20))), f(rng.uniform(.3, 1))) for _ in range(2)]
    for pos, inten in lights:
        i = scene.add_sphere(pos, 0.2)
        scene.set_light(i)
        scene.set_intensity(i, inten)
    n = int(rng.choice([70, 90, 130]))
    for k in range(n):                      # enough spheres in a row to be clustered
        r = f(rng.choice([1e-3, 0.05, 0.4, 1.0, 6.0]))
        c = (f(rng.uniform(-25, 25)), f(rng.uniform(3, 400 if k % 7 == 0 else 60)), f(rng.uniform(0, 8)))
        i = scene.add_sphere(c, r)
        scene.set_color(i, [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)][k % 4])
        if k % 3 == 0:
            scene.set_reflective(i, f(rng.choice([0.3, 1.0])))
            scene.set_diffuse(i, f(rng.choice([0.0, 0.6])))
        if k % 11 == 0:                     # an exact duplicate: equal distances, the lower index must win
            j = scene.add_sphere(c, r)
            scene.set_color(j, (1, 1, 0))
    for k in range(6):                      # skewed finite planes through the axis constructor
        o = (f(rng.uniform(-10, 10)), f(rng.uniform(5, 40)), f(rng.uniform(0, 6)))
        nrm = tuple(f(v) for v in rng.uniform(-1, 1, 3))
        hor = tuple(f(v) for v in rng.uniform(-1, 1, 3))
        i = scene.add_finite_plane_axes(o, nrm, hor, f(rng.uniform(1, 9)), f(rng.uniform(1, 9)))
        scene.set_reflective(i, 0.5)
    for k in range(2):                      # boxes out of three-corner rectangles (axis-aligned class runs)
        o = np.float32([rng.uniform(-12, 12), rng.uniform(6, 30), 0])
        dims = np.float32(rng.uniform(0.5, 5, 3))
        c = [o.copy() for _ in range(8)]
        c[1][0] += dims[0]; c[2][1] += dims[1]; c[3][2] += dims[2]
        c[4][0] += dims[0]; c[4][1] += dims[1]; c[5][0] += dims[0]; c[5][2] += dims[2]
        c[6][1] += dims[1]; c[6][2] += dims[2]; c[7] = o + dims
        for a, b, d in ((0, 3, 2), (0, 3, 1), (0, 1, 2), (7, 4, 6), (7, 4, 5), (7, 5, 6)):
            i = scene.add_finite_plane_corners(tuple(map(float, c[a])), tuple(map(float, c[b])), tuple(map(float, c[d])))
            scene.set_color(i, (0.2, 0.2, 0.0))
            if k == 1:
                scene.set_reflective(i, 0.5)
    i = scene.add_infinite_plane((0, 0, 0), (0, 0, 1), (1, 0, 0))
    scene.set_checkerboard(i, (1, 1, 1), (0, 0, 0), 3.0, 3.0)
    scene.set_reflective(i, 0.5)
    scene.set_diffuse(i, 0.5)
    i = scene.add_infinite_plane((0, 0, 25), (0, 0, -1), (1, 0, 0))
    scene.set_reflective(i, 0.4)
    mode = seed % 3
    if mode == 0:
        scene.set_object_indices(0, 1)
    elif mode == 1:
        scene.set_object_indices(1, 3)      # partial shadow range: clustering must stay exact (or switch itself off)
    scene.camera_two_mirrors()
    return scene


@pytest.mark.parametrize("seed", range(20, 32))
def test_adversarial_scenes(oracle, seed):
    host = _adversarial(HostScene.empty(), seed)
    orc = _adversarial(oracle.OracleScene(), seed)
    assert_same(Renderer(host).render(96, 64, 5), orc.render(96, 64, 5), f"adversarial seed {seed}")


@pytest.mark.parametrize("leaf", [0, 4, 8, 16, 32])
def test_cluster_parameters_do_not_change_results(oracle, leaf):
    host = _adversarial(HostScene.empty(), 77)
    r = Renderer(host)
    r.set_option("cluster_leaf", leaf)
    assert_same(r.render(80, 48, 4), _adversarial(oracle.OracleScene(), 77).render(80, 48, 4), f"leaf {leaf}")


def test_aa_fast_path_can_be_switched_off(oracle):
    r = Renderer(HostScene.builtin())
    r.set_option("aa_planes", 0)
    assert_same(r.render(120, 90, 4), oracle.OracleScene.builtin().render(120, 90, 4), "aa_planes=0")


@pytest.mark.parametrize("seed", range(40, 70))
def test_more_random_scenes(oracle, seed):
    from scene_gen import build_random
    kw = dict(n_spheres=int(4 + seed % 9), n_finite=int(seed % 7), n_infinite=int(seed % 3), n_lights=1 + seed % 3,
              shadows=(seed % 4 != 0))
    host = build_random(HostScene.empty(), seed, **kw)
    orc = build_random(oracle.OracleScene(), seed, **kw)
    assert_same(Renderer(host).render(64, 48, 6), orc.render(64, 48, 6), f"seed {seed}")


def test_degenerate_geometry(oracle):
    """Zero and negative radii, a plane with a zero normal (its normalisation divides by
    zero), a rectangle of zero width, the camera inside a mirror sphere: whatever the
    reference's arithmetic makes of them, the kernel must make the same."""
    host, orc = HostScene.empty(), oracle.OracleScene()
    for s in (host, orc):
        i = s.add_sphere((3.0, 5.0, 8.0), 0.15)
        s.set_light(i)
        s.add_sphere((0.0, 6.0, 1.0), 0.0)
        s.add_sphere((1.0, 6.0, 1.0), -1.0)
        s.add_infinite_plane((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 0.0, 0.0))
        s.add_finite_plane_axes((0.0, 8.0, 0.0), (0.0, -1.0, 0.0), (1.0, 0.0, 0.0), 0.0, 3.0)
        i = s.add_infinite_plane((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
        s.set_reflective(i, 0.5)
        i = s.add_sphere((0.0, -1.0, 2.5), 3.0)
        s.set_reflective(i, 1.0)
        s.set_object_indices(0, 1)
        s.camera_two_mirrors()
    assert_same(Renderer(host).render(48, 40, 4), orc.render(48, 40, 4), "degenerate geometry")


@pytest.mark.parametrize("seed,n", [(1, 64), (2, 120), (3, 200), (4, 333), (5, 90), (6, 150)])
def test_random_sphere_fields(oracle, seed, n):
    """Clustered sphere runs of mixed sizes seen to the horizon (far hit points)."""
    from scene_gen import build_sphere_field
    host = build_sphere_field(HostScene.empty(), seed, n_spheres=n)
    orc = build_sphere_field(oracle.OracleScene(), seed, n_spheres=n)
    r = Renderer(host)
    assert_same(r.render(96, 80, 4), orc.render(96, 80, 4), f"field {seed}")
    # the rows around the horizon at a finer pitch: rows 2040..2055 of a 4096-row image, 48 columns
    W, H = 48, 4096
    img = r.render(W, H, 3)
    want = orc.render(W, H, 3)
    assert_same(img[:, 2040:2056], want[:, 2040:2056], f"field {seed} horizon rows")
    assert_same(img, want, f"field {seed} tall strip")


@pytest.mark.parametrize("svox", [0, 800, 4096, 300000])
@pytest.mark.parametrize("name,W,H,depth", [("grid16", 96, 160, 8), ("grid32", 128, 96, 4), ("grid9", 50, 120, 3)])
def test_shadow_voxels_on_the_sphere_grids(oracle, name, W, H, depth, svox):
    """rt_set_option("svox", n): the per-voxel, per-light masks of leaves that can block (SHADOW VOXELS, csrc/rt_tables.h) with
    no table, the coarsest grid there is (a core of one or two cells per axis: nearly everything lies in the tail cells), the
    automatic budget and a very fine grid -- forced on from four leaves (automatic: from 24)."""
    r = Renderer(HostScene.named(name))
    r.set_option("svox", svox)
    assert_same(r.render(W, H, depth), oracle.OracleScene.named(name).render(W, H, depth), f"{name} svox={svox}")


def test_shadow_voxels_take_candidates_away(oracle):
    """What the table is for (counting build, 1 024-sphere grid, 512^2): the leaves a shadow scan's bundle cull leaves as candidates
    -- each a box test for the whole wavefront -- drop to less than 0.55 of what they were (at 512^2, whose tiles span twice the scene
    of a 1024^2 frame's: 11.9 -> 5.7 per scan; 1024^2: 8.9 -> 4.4, the review's mark being five), with the same image, the same rays and
    no more leaves that some lane needs."""
    want = oracle.OracleScene.named("grid32").render(512, 512, 4)
    r = Renderer(HostScene.named("grid32"))
    img_on, on = r.render_stats(512, 512, 4)
    r.set_option("svox", 0)
    img_off, off = r.render_stats(512, 512, 4)
    assert_same(img_on, want, "grid32 counting build, table on")
    assert_same(img_off, want, "grid32 counting build, table off")
    assert on["shadow_rays"] == off["shadow_rays"] and on["wave_shadow_scans"] == off["wave_shadow_scans"]
    per_scan_on = on["shadow_candidates"] / on["wave_shadow_scans"]
    per_scan_off = off["shadow_candidates"] / off["wave_shadow_scans"]
    assert per_scan_off > 7.0 and per_scan_on < 0.55 * per_scan_off and per_scan_on < 6.5, (per_scan_on, per_scan_off)
    assert on["wave_box_tests"] < 0.75 * off["wave_box_tests"]
    assert on["shadow_leaves_union"] <= off["shadow_leaves_union"]


@pytest.mark.parametrize("seed,n", [(1, 64), (3, 200), (4, 333), (11, 700), (12, 1000)])
def test_shadow_voxels_on_random_sphere_fields(oracle, seed, n):
    """The same on clustered fields of spheres of mixed sizes with the lights INSIDE the field's box, seen to the horizon: shading
    points from inside the leaves' boxes to tens of thousands of units away (the tail cells, and beyond the last one), every
    slack of the float sphere test at work.  Whole small frames and the rows around the horizon of a
    4 096-row strip, table forced on / automatic / off."""
    from scene_gen import build_sphere_field
    host = build_sphere_field(HostScene.empty(), seed, n_spheres=n)
    orc = build_sphere_field(oracle.OracleScene(), seed, n_spheres=n)
    want_small = orc.render(96, 80, 4)
    W, H = 16, 4096
    want_tall = orc.render(W, H, 3, 0, W)
    r = Renderer(host)
    for svox in (1500, -1, 0):
        r.set_option("svox", svox)
        assert_same(r.render(96, 80, 4), want_small, f"field {seed} svox={svox}")
        assert_same(r.render(W, H, 3), want_tall, f"field {seed} tall strip svox={svox}")


def test_counting_build_matches_and_counts(oracle):
    r = Renderer(HostScene.builtin())
    img, st = r.render_stats(64, 64, 3)
    assert_same(img, oracle.OracleScene.builtin().render(64, 64, 3), "counting build")
    orc = oracle.OracleScene.builtin()
    orc.render(64, 64, 3)
    c = oracle.OracleScene.counters()
    assert st["nearest_rays"] == c.nearest_rays          # same rays traced as the reference restatement
    assert st["shadow_rays"] == c.shadow_rays


# ------------------------------------------------ round 2: culls against the plain scan
def _nested_spheres(scene, swap=False):
    """The camera (setSceneTwoMirrors: eye (0,-1,2.5)) inside two nested spheres.  A ray that
    starts inside a sphere reports root1 = v - sqrt(d^2) < 0 (src/SceneSphere.cpp:136-140), so
    the OUTER sphere is the nearest hit whatever the Scene order; with >= 8 items and >= 4
    candidates the nearest-first route of the kernel runs, and it must not stop after the first
    sphere whose box contains the origins."""
    i = scene.add_sphere((3.0, 5.0, 8.0), 0.15)
    scene.set_light(i)
    radii = (9.0, 4.0) if swap else (4.0, 9.0)
    for k, r in enumerate(radii):
        i = scene.add_sphere((0.25, -0.5, 2.0), r)
        scene.set_color(i, [(1, 0, 0), (0, 0, 1)][k])
        scene.set_reflective(i, 0.5)
        scene.set_diffuse(i, 0.5)
    for k in range(7):
        i = scene.add_sphere((-3.0 + k, 1.5 + 0.25 * k, 2.0 + 0.1 * k), 0.4)
        scene.set_color(i, (0, 1, 0))
        if k % 2:
            scene.set_reflective(i, 1.0)
    i = scene.add_infinite_plane((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
    scene.set_reflective(i, 0.5)
    i = scene.add_infinite_plane((0.0, 0.0, 6.0), (0.0, 0.0, -1.0), (1.0, 0.0, 0.0))
    scene.set_specular(i, 0.5)
    scene.set_object_indices(0, 1)
    scene.camera_two_mirrors()
    return scene


@pytest.mark.parametrize("swap", [False, True])
def test_camera_inside_nested_spheres(oracle, swap):
    host = _nested_spheres(HostScene.empty(), swap)
    orc = _nested_spheres(oracle.OracleScene(), swap)
    want = orc.render(72, 64, 5)
    r = Renderer(host)
    assert_same(r.render(72, 64, 5), want, f"nested spheres, swap={swap}")
    r.set_option("cull", 0)
    assert_same(r.render(72, 64, 5), want, f"nested spheres, plain scan, swap={swap}")


def test_inside_a_clustered_sphere_field(oracle):
    """Negative-distance hits inside a clustered run: overlapping big spheres around the camera."""
    def build(scene):
        rng = np.random.RandomState(7)
        i = scene.add_sphere((3.0, 5.0, 8.0), 0.15)
        scene.set_light(i)
        for k in range(80):
            big = k % 9 == 0
            c = (float(np.float32(rng.uniform(-2, 2))), float(np.float32(rng.uniform(-2, 6))), float(np.float32(rng.uniform(1, 4))))
            i = scene.add_sphere(c, float(np.float32(rng.uniform(6, 12) if big else rng.uniform(0.2, 0.8))))
            scene.set_color(i, [(1, 0, 0), (0, 1, 0), (0, 0, 1)][k % 3])
            if k % 2:
                scene.set_reflective(i, 0.5)
        scene.add_infinite_plane((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
        scene.set_object_indices(0, 1)
        scene.camera_two_mirrors()
        return scene
    want = build(oracle.OracleScene()).render(64, 64, 4)
    r = Renderer(build(HostScene.empty()))
    assert_same(r.render(64, 64, 4), want, "inside a clustered field")
    r.set_option("cull", 0)
    assert_same(r.render(64, 64, 4), want, "inside a clustered field, plain scan")


@pytest.mark.parametrize("name,W,H,depth", [("builtin", 150, 130, 4), ("grid16", 96, 96, 8), ("grid32", 64, 96, 4), ("twomirrors", 32, 32, 5)])
def test_plain_scan_option(oracle, name, W, H, depth):
    """rt_set_option("cull", 0): no bundle culls, no nearest-first exit, no sphere clustering, no
    axis-aligned route -- the in-order scans of src/RayTracer.cpp:50-89, 709-739 as they stand."""
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.set_option("cull", 0)
    assert_same(r.render(W, H, depth), want, f"{name}, cull 0")
    r.set_option("cull", 1)
    assert_same(r.render(W, H, depth), want, f"{name}, cull 1 again")


def _frames_equal_on_device(a, b):
    import torch
    return bool(torch.equal(a.view(torch.int32), b.view(torch.int32)))


@pytest.mark.parametrize("name,depth,cols", [
    ("grid32", 4, [3, 2047, 2048, 4001]),            # configs[2]
    ("grid16", 8, [9, 1531, 2050, 3777]),            # configs[4]
])
def test_4096_sphere_grid_every_pixel_culls_vs_plain_scan(oracle, name, depth, cols):
    """Full 4096 x 4096 frames of the sphere-grid configs: the default kernel (bundle culls,
    nearest-first exit, clustered runs) against the plain in-order scan on EVERY pixel, and the
    plain frame against the oracle on whole columns (horizon rows included)."""
    import torch
    W = H = 4096
    r = Renderer(HostScene.named(name))
    stream = torch.cuda.current_stream().cuda_stream
    fast = torch.empty((W, H, 3), dtype=torch.float32, device="cuda:0")
    plain = torch.empty((W, H, 3), dtype=torch.float32, device="cuda:0")
    r.render_device(W, H, depth, 0, W, fast.data_ptr(), stream)
    r.set_option("cull", 0)
    r.render_device(W, H, depth, 0, W, plain.data_ptr(), stream)
    torch.cuda.synchronize()
    if not _frames_equal_on_device(fast, plain):
        bad = (fast.view(torch.int32) != plain.view(torch.int32)).any(dim=-1).nonzero()
        raise AssertionError(f"{name}: {len(bad)} pixels differ between the default kernel and the plain scan, first {bad[0].tolist()}")
    for x0, want in _oracle_columns(oracle, lambda: oracle.OracleScene.named(name), W, H, depth, cols).items():
        assert_same(plain[x0:x0 + 1].cpu().numpy(), want, f"{name} plain scan, column {x0}")


def test_8192_builtin_as_eight_strips(oracle):
    """configs[3]: the 8192 x 8192 frame as eight 1024-column x-strips, the per-rank calls of the
    static partition (src/RayTracer.cpp:904-923 with CORE_NUM = 8) through rt_render_device."""
    import torch
    W = H = 8192
    depth = 4
    r = Renderer(HostScene.builtin())
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.empty((W, H, 3), dtype=torch.float32, device="cuda:0")
    r.render_device(W, H, depth, 0, W, full.data_ptr(), stream)
    strip = torch.empty((1024, H, 3), dtype=torch.float32, device="cuda:0")
    for g in range(8):
        strip.fill_(-1.0)
        r.render_device(W, H, depth, g * 1024, (g + 1) * 1024, strip.data_ptr(), stream)
        torch.cuda.synchronize()
        assert _frames_equal_on_device(strip, full[g * 1024:(g + 1) * 1024]), f"strip {g} differs from the full render"
    want = np.fromfile(os.path.join(GOLDEN, "b64d4.f32"), dtype=np.float32).reshape(64, 64, 3)
    assert_same(full[::128, ::128].contiguous().cpu().numpy(), want, "stride-128 subsample")   # (float)(128k)/8192 == (float)k/64
    cols = [1, 513, 1023, 1025, 2047, 2049, 3071, 3073, 4095, 4097, 5119, 5121, 6143, 6145, 7167, 7169, 8191]
    for x0, got in _oracle_columns(oracle, lambda: oracle.OracleScene.builtin(), W, H, depth, cols).items():
        assert_same(full[x0:x0 + 1].cpu().numpy(), got, f"column {x0}")
    # and every one of the 67 108 864 pixels against the oracle (five seconds of the box's 16 cores)
    host_image = full.cpu().numpy()
    del full, strip
    _assert_every_pixel(oracle, host_image, lambda: oracle.OracleScene.builtin(), depth, "8192^2 built-in", size=8192)


def test_8192_grid32_as_eight_strips(oracle):
    """configs[3] on the sphere-grid scene (C4 for both scenes): the 8192 x 8192 frame of the 1 024-sphere grid
    as eight 1024-column x-strips -- every strip is a launch with HELP, HEAVY tiles and tile priorities on, the frame one
    without -- against the full render on the device, and whole columns of it (strip edges, the image's edges) against the
    oracle."""
    import torch
    W = H = 8192
    depth = 4
    r = Renderer(HostScene.named("grid32"))
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.empty((W, H, 3), dtype=torch.float32, device="cuda:0")
    r.render_device(W, H, depth, 0, W, full.data_ptr(), stream)
    strip = torch.empty((1024, H, 3), dtype=torch.float32, device="cuda:0")
    for g in range(8):
        strip.fill_(-1.0)
        r.render_device(W, H, depth, g * 1024, (g + 1) * 1024, strip.data_ptr(), stream)
        torch.cuda.synchronize()
        assert _frames_equal_on_device(strip, full[g * 1024:(g + 1) * 1024]), f"strip {g} differs from the full render"
    # (the survey's 64 x 64 fixture is the NO-shadow grid: the stride-128 subsample of this frame has no digest to meet)
    cols = [0, 1023, 1024, 2047, 4095, 4096, 4097, 7168, 8191]
    for x0, got in _oracle_columns(oracle, lambda: oracle.OracleScene.named("grid32"), W, H, depth, cols).items():
        assert_same(full[x0:x0 + 1].cpu().numpy(), got, f"column {x0}")
    r.timing()                                # (a HELP wait that timed out would be reported here)


@pytest.mark.parametrize("seed", [7001, 7002, 7003, 7004, 7005, 7006])
def test_far_origin_grazing_rays(oracle, seed):
    """Rays that arrive at tilted rectangles near the origin from 1e4 ... 6e4 units away, grazing them (scene_gen.
    build_far_grazing: shadow rays of ground hits at the horizon, camera rays sent back by a far mirror wall): the tight
    plane boxes of the culls (csrc/rt_capi.hip box_item(), RT_PLANE_SLACK) must still hold every hit the reference's float
    arithmetic reports.  Strips 8 columns wide and 16 384 / 32 768 rows tall: the rows around the middle are the far ones.
    Compared with the oracle and -- every pixel -- with the plain in-order scans."""
    from scene_gen import build_far_grazing
    host, orc = build_far_grazing(HostScene.empty(), seed), build_far_grazing(oracle.OracleScene(), seed)
    H = 16384 if seed % 2 else 32768
    r = Renderer(host)
    got = r.render(8, H, 3)
    assert_same(got, orc.render(8, H, 3), f"far grazing {seed}")
    r.set_option("tight_planes", 0)
    assert_same(r.render(8, H, 3), got, f"far grazing {seed}, sphere padding for planes")
    r.set_option("cull", 0)
    assert_same(r.render(8, H, 3), got, f"far grazing {seed}, plain scans")


@pytest.mark.parametrize("first", [5000, 5025, 5050, 5075])
def test_fuzz_slice(oracle, first):
    """A bounded slice (100 seeds in all) of scripts/fuzz_gpu.py's randomised sweep."""
    from scene_gen import build_random, build_sphere_field
    for seed in range(first, first + 25):
        rng = np.random.RandomState(seed)
        if seed % 4 == 0:
            n = int(rng.choice([64, 80, 130, 260]))
            spread = float(rng.choice([30.0, 60.0, 200.0]))
            mk = lambda s: build_sphere_field(s, seed, n_spheres=n, spread=spread)
            W, H, depth = 40, int(rng.choice([64, 512, 2048])), int(rng.randint(1, 6))
        else:
            kw = dict(n_spheres=int(rng.randint(0, 40)), n_finite=int(rng.randint(0, 12)), n_infinite=int(rng.randint(0, 3)),
                      n_lights=int(rng.randint(1, 4)), shadows=bool(rng.rand() < 0.8))
            mk = lambda s: build_random(s, seed, **kw)
            W, H, depth = int(rng.randint(1, 90)), int(rng.randint(1, 90)), int(rng.randint(0, 9))
        r = Renderer(mk(HostScene.empty()))
        if seed % 3 == 0:
            r.set_option("tile_z", int(2 ** rng.randint(0, 7)))
        assert_same(r.render(W, H, depth), mk(oracle.OracleScene()).render(W, H, depth), f"fuzz seed {seed}")


def test_failed_option_leaves_the_handle_usable(oracle):
    """An option the scene cannot be re-packed with is refused and changes nothing."""
    from tilecoderaytracer_amd import RtError
    want = oracle.OracleScene.two_mirrors().render(24, 24, 3)
    r = Renderer(HostScene.two_mirrors())
    for key, value in (("cluster_leaf", 0), ("cluster_leaf", 4), ("cull", 0)):
        try:
            r.set_option(key, value)
        except RtError:
            pass
        assert_same(r.render(24, 24, 3), want, f"after {key}={value}")


# ------------------------------------------------ shared shadow scans: HELP and HEAVY tiles on awkward scenes
# (round 2's second launch for deferred tiles is gone: HEAVY tiles do its job inside the one launch)
@pytest.mark.parametrize("heavy,block,tile_z", [(0, 0, 0), (1, 128, 0), (6, 256, 4), (2, 64, 16), (5, 192, 1), (9, 0, 64), (400, 0, 0), (3, 0, 8)])
def test_shared_scans_on_a_dense_sphere_field(oracle, heavy, block, tile_z):
    """A field of 150 overlapping spheres of very different sizes seen to the horizon: every tile shape and workgroup
    size, help from two candidate leaves on, the HEAVY band off / one row / the whole image."""
    from scene_gen import build_sphere_field
    host = build_sphere_field(HostScene.empty(), 11, n_spheres=150)
    orc = build_sphere_field(oracle.OracleScene(), 11, n_spheres=150)
    r = Renderer(host)
    r.set_option("help", 2)
    r.set_option("heavy", heavy)
    if block:
        r.set_option("block_threads", block)
    if tile_z:
        r.set_option("tile_z", tile_z)
    want = orc.render(72, 333, 4)
    assert_same(r.render(72, 333, 4), want, f"heavy {heavy} block {block} tile_z {tile_z}")
    assert_same(r.render(72, 333, 4, 30, 71), want[30:71], f"heavy {heavy} block {block} tile_z {tile_z}, strip")


def test_shared_scans_inside_a_clustered_sphere_field(oracle):
    """Negative-distance hits (ray origins inside spheres) with the shadow scans shared (HELP, HEAVY tiles)."""
    def build(scene):
        rng = np.random.RandomState(7)
        i = scene.add_sphere((3.0, 5.0, 8.0), 0.15)
        scene.set_light(i)
        for k in range(80):
            big = k % 9 == 0
            c = (float(np.float32(rng.uniform(-2, 2))), float(np.float32(rng.uniform(-2, 6))), float(np.float32(rng.uniform(1, 4))))
            i = scene.add_sphere(c, float(np.float32(rng.uniform(6, 12) if big else rng.uniform(0.2, 0.8))))
            if k % 2:
                scene.set_reflective(i, 0.5)
        scene.add_infinite_plane((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
        scene.set_object_indices(0, 1)
        scene.camera_two_mirrors()
        return scene
    want = build(oracle.OracleScene()).render(64, 64, 4)
    r = Renderer(build(HostScene.empty()))
    r.set_option("help", 2)
    for heavy in (0, 1, 100):
        r.set_option("heavy", heavy)
        assert_same(r.render(64, 64, 4), want, f"inside a clustered field, heavy {heavy}")


# ------------------------------------------------ round 2: HELP (wavefronts out of tiles serve their workgroup's shadow scans)
@pytest.mark.parametrize("help_leaves", [2, 5, 16, 64])
@pytest.mark.parametrize("block", [128, 192, 256])
@pytest.mark.parametrize("name,W,H,depth", [("grid16", 96, 80, 8), ("grid32", 128, 64, 4), ("grid9", 50, 120, 3)])
def test_help_does_not_change_results(oracle, name, W, H, depth, block, help_leaves):
    """rt_set_option("help", n): a shadow scan left with n or more candidate leaves is shared with the workgroup's
    wavefronts that are out of tiles.  Small images on a grid of one tile per wavefront: every workgroup has idle
    wavefronts while others still scan.  Same pixels as the oracle, whole image and strip."""
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.set_option("help", help_leaves)
    r.set_option("block_threads", block)
    for _ in range(2):                            # who helps whom depends on timing: twice
        assert_same(r.render(W, H, depth), want, f"{name} help {help_leaves}, block {block}")
    assert_same(r.render(W, H, depth, 5, W - 9), want[5:W - 9], f"{name} help {help_leaves}, block {block}, strip")


@pytest.mark.parametrize("seed", [21, 24, 27, 30])
def test_help_on_adversarial_scenes(oracle, seed):
    host = _adversarial(HostScene.empty(), seed)
    orc = _adversarial(oracle.OracleScene(), seed)
    r = Renderer(host)
    r.set_option("help", [2, 3, 9, 1][seed % 4])
    assert_same(r.render(96, 64, 5), orc.render(96, 64, 5), f"adversarial seed {seed}, help")


def test_help_on_and_off_render_the_same_strip_of_a_large_frame():
    """4096^2 frame of the 1 024-sphere grid, the strip of GPU 3 of 8: help on, help from 2 leaves on, help off, automatic (on: a strip)."""
    imgs = []
    for value in (1, 2, 0, -1):
        r = Renderer(HostScene.named("grid32"))
        r.set_option("help", value)
        imgs.append(r.render(4096, 4096, 4, 1536, 2048))
    assert imgs[0].shape == (512, 4096, 3)
    for img, what in ((imgs[1], "help 2"), (imgs[2], "help off"), (imgs[3], "help automatic")):
        assert np.array_equal(imgs[0].view(np.uint32), img.view(np.uint32)), what


@pytest.mark.parametrize("heavy", [-1, 1, 3, 40])
@pytest.mark.parametrize("name,W,H,depth,x0,x1", [("grid16", 96, 160, 8, 0, 96), ("grid32", 128, 96, 4, 16, 100), ("grid9", 50, 120, 3, 0, 50)])
def test_heavy_tiles_do_not_change_results(oracle, name, W, H, depth, x0, x1, heavy):
    """rt_set_option("heavy", k): the band of tile rows along the horizon line is rendered first, one tile per
    workgroup, its shadow scans shared with the workgroup's other wavefronts from the first scan on.  Scheduling
    only: same pixels as the oracle, with the band one row wide, a few rows wide, wider than the image, automatic."""
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.set_option("heavy", heavy)
    r.set_option("help", 2)
    for _ in range(2):
        assert_same(r.render(W, H, depth, x0, x1), want[x0:x1], f"{name} heavy {heavy}")


@pytest.mark.parametrize("name,W,H,depth", [("builtin", 70, 90, 6), ("grid16", 96, 64, 8), ("twomirrors", 64, 64, 12)])
def test_tile_priority_does_not_change_results(oracle, name, W, H, depth):
    """rt_set_option("tile_prio", k): wavefronts raise their priority with their tile's bounce level (scheduling only)."""
    from tilecoderaytracer_amd import RtError
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    for prio in (1, 0, -1):
        r.set_option("tile_prio", prio)
        assert_same(r.render(W, H, depth), want, f"{name} tile_prio {prio}")
        assert_same(r.render(W, H, depth, 8, 24), want[8:24], f"{name} tile_prio {prio}, strip")
    with pytest.raises(RtError):
        r.set_option("tile_prio", 2)


def test_heavy_tiles_on_a_tilted_horizon(oracle):
    """A camera rolled about its viewing axis: the horizon line is slanted, the band follows it column by column."""
    from scene_gen import build_sphere_field
    host = build_sphere_field(HostScene.empty(), 3, n_spheres=90)
    orc = build_sphere_field(oracle.OracleScene(), 3, n_spheres=90)
    hc = host.camera.contents
    h = np.array(list(hc.vector_horizontal), dtype=np.float32)
    v = np.array(list(hc.vector_vertical), dtype=np.float32)
    ca, sa = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    h2, v2 = ca * h + sa * v, ca * v - sa * h
    so = np.array(list(hc.screen_origin), dtype=np.float32) + np.float32(0.35) * v       # keep the horizon inside the image
    for k in range(3):
        hc.vector_horizontal[k], hc.vector_vertical[k], hc.screen_origin[k] = float(h2[k]), float(v2[k]), float(so[k])
    orc.cam.vector_horizontal = type(orc.cam.vector_horizontal)(float(h2[0]), float(h2[1]), float(h2[2]))
    orc.cam.vector_vertical = type(orc.cam.vector_vertical)(float(v2[0]), float(v2[1]), float(v2[2]))
    orc.cam.screen_origin = type(orc.cam.screen_origin)(float(so[0]), float(so[1]), float(so[2]))
    want = orc.render(144, 128, 3)
    r = Renderer(host)
    r.set_option("help", 2)
    for heavy in (2, 5, -1):
        r.set_option("heavy", heavy)
        assert_same(r.render(144, 128, 3), want, f"tilted horizon, heavy {heavy}")
        assert_same(r.render(144, 128, 3, 40, 101), want[40:101], f"tilted horizon, heavy {heavy}, strip")


def test_timeline_is_a_diagnostic_build_feature(oracle):
    """The per-tile timeline (scripts/timeline_gpu.py) costs the render kernels registers, so only a diagnostic
    build records it (make variant DEFS=-DRT_TIMELINE=1); the product library says so instead of ignoring the option."""
    from tilecoderaytracer_amd import RtError, capi
    r = Renderer(HostScene.named("grid9"))
    r.set_option("timeline", 0)
    try:
        r.set_option("timeline", 1)
    except RtError as e:
        assert e.code == capi.RT_ERR_INVALID and "diagnostic" in e.message
        return
    # a diagnostic build: every tile has a record, rendered once, and the image is unaffected
    want = oracle.OracleScene.named("grid9").render(160, 96, 3)
    assert_same(r.render(160, 96, 3), want, "with the timeline on")
    rec = r.timeline(0, 160, 96)
    assert (rec[..., 0] > 0).all() and (rec[..., 1] >= rec[..., 0]).all()


def test_help_timeout_path_is_exact_and_reported(oracle):
    """rt_set_option("help_spin_limit", -1): every wait of an owner for its helpers counts as timed out.  The owner
    then tests every leaf of the scan itself (an OR: the pixels cannot change), its workgroup stops helping, and
    the host is told once: RT_ERR_HIP from the call that finds the flag, with the image delivered."""
    import ctypes as C
    from tilecoderaytracer_amd import capi
    lib = capi.load_library()
    name, W, H, depth = "grid16", 96, 80, 8
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.set_option("help", 2)
    r.set_option("block_threads", 256)
    r.set_option("help_spin_limit", -1)
    out = np.zeros((W, H, 3), dtype=np.float32)
    rc = lib.rt_render(r._scene, r._cam, W, H, 0, W, depth, out.ctypes.data)
    assert rc == capi.RT_ERR_HIP, "the timeout must be reported"
    assert b"HELP" in lib.rt_last_error()
    assert_same(out, want, "image of the launch whose HELP waits timed out")
    r.set_option("help_spin_limit", 1 << 22)
    assert_same(r.render(W, H, depth), want, "the handle is usable afterwards, and reports nothing")


@pytest.mark.parametrize("name,W,H,depth,x0,x1", [("builtin", 150, 200, 5, 0, 150), ("grid16", 96, 160, 8, 10, 90), ("grid32", 128, 96, 4, 0, 128),
                                                  ("twomirrors", 64, 64, 12, 0, 64)])
def test_learned_tile_order_does_not_change_results(oracle, name, W, H, depth, x0, x1):
    """rt_learn_tile_order: the macro rows of later launches of the same shape come out most expensive first (by one frame of
    the counting build).  Scheduling only: same pixels; another shape renders by the rule; option "learned_order" 0 forgets."""
    from tilecoderaytracer_amd import RtError
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.learn_tile_order(W, H, depth, x0, x1)
    for _ in range(2):
        assert_same(r.render(W, H, depth, x0, x1), want[x0:x1], f"{name}, learned order")
    assert_same(r.render(W, H, depth, x0, x1 - 1), want[x0:x1 - 1], f"{name}, another shape after learning")
    r.set_option("tile_z", 8)
    assert_same(r.render(W, H, depth, x0, x1), want[x0:x1], f"{name}, another tile shape after learning")
    r.set_option("learned_order", 0)
    assert_same(r.render(W, H, depth, x0, x1), want[x0:x1], f"{name}, order forgotten")
    with pytest.raises(RtError):
        r.set_option("learned_order", 1)
    with pytest.raises(RtError):
        r.learn_tile_order(W, H, depth, 5, 5)


def test_help_option_range():
    from tilecoderaytracer_amd import RtError
    r = Renderer(HostScene.named("grid9"))
    for bad in (-2, 65):
        with pytest.raises(RtError):
            r.set_option("help", bad)
    r.set_option("help", -1)          # automatic (the default): on for strips, off for whole frames
    assert r.render(16, 16, 2).shape == (16, 16, 3)


# ------------------------------------------------ round 2: scenes larger than LDS (tables in global memory)
def _mixed_scene(scene, n_objects, seed=3):
    """Alternating spheres and finite planes (no runs: nothing clusters, every object is a plain
    item with its full record) up to the reference's Scene capacity (MAX_OBJECT_COUNT 4000, of which
    addObject fills 3 999, src/Scene.h:8, src/Scene.cpp:470-479)."""
    rng = np.random.RandomState(seed)
    f = lambda x: float(np.float32(x))
    i = scene.add_sphere((3.0, 5.0, 9.0), 0.15)
    scene.set_light(i)
    for k in range(n_objects - 1):
        c = (f(rng.uniform(-9, 9)), f(rng.uniform(3, 40)), f(rng.uniform(0, 7)))
        if k % 2 == 0:
            i = scene.add_sphere(c, f(rng.uniform(0.05, 0.35)))
        elif k % 4 == 1:
            i = scene.add_finite_plane_axes(c, tuple(f(v) for v in rng.uniform(-1, 1, 3)), tuple(f(v) for v in rng.uniform(-1, 1, 3)),
                                            f(rng.uniform(0.2, 0.8)), f(rng.uniform(0.2, 0.8)))
        else:
            i = scene.add_finite_plane_corners(c, (c[0], c[1], f(c[2] + 0.5)), (f(c[0] + 0.5), c[1], c[2]))   # axis-aligned
        scene.set_color(i, [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0)][k % 4])
        if k % 5 == 0:
            scene.set_reflective(i, 0.5)
    scene.set_object_indices(0, 1)
    scene.camera_two_mirrors()
    return scene


def test_3999_object_mixed_scene(oracle):
    host = _mixed_scene(HostScene.empty(), 3999)
    assert host.object_count == 3999
    r = Renderer(host)
    want = _mixed_scene(oracle.OracleScene(), 3999).render(48, 40, 3)
    assert_same(r.render(48, 40, 3), want, "3 999 mixed objects")
    li = r.launch_info()
    assert li.scene_lds_bytes == 0                      # tables of ~500 KB: read from global memory
    r.set_option("cull", 0)
    assert_same(r.render(48, 40, 3, 5, 29), want[5:29], "3 999 mixed objects, plain scan, strip")


@pytest.mark.parametrize("tables", [1, 2])
@pytest.mark.parametrize("name,W,H,depth", [("builtin", 100, 90, 4), ("grid16", 64, 64, 8), ("twomirrors", 32, 32, 6)])
def test_tables_in_lds_or_global_memory(oracle, name, W, H, depth, tables):
    want = oracle.OracleScene.named(name).render(W, H, depth)
    r = Renderer(HostScene.named(name))
    r.set_option("tables", tables)
    assert_same(r.render(W, H, depth), want, f"{name}, tables={tables}")
    assert (r.launch_info().scene_lds_bytes == 0) == (tables == 2)


def test_two_mirrors_512_depth_50(oracle):
    """The reference's SCENE 2 (3 920 objects, facing mirrors) at 512 x 512, MAX_RECURSION_LEVEL 50,
    every pixel; then the same with the plain scans (no clusters: 313 KB of tables, global memory)."""
    import ctypes as C
    W = H = 512
    scene = oracle.OracleScene.two_mirrors()
    want = np.empty((W, H, 3), np.float32)
    starts = (C.c_int * (W // 8))(*range(0, W, 8))
    sec = C.c_double()
    assert oracle.LIB.orc_render_static_partition(scene.h, C.byref(scene.cam), W, H, 50, starts, W // 8, 8, 8, None,
                                                  want.ctypes.data, C.byref(sec)) == 0
    r = Renderer(HostScene.two_mirrors())
    assert_same(r.render(W, H, 50), want, "two mirrors 512^2 d50")
    r.set_option("cull", 0)
    assert_same(r.render(W, H, 50, 200, 264), want[200:264], "two mirrors 512^2 d50, plain scan, columns 200:264")
    assert r.launch_info().scene_lds_bytes == 0


def test_tables_option_that_does_not_fit_is_refused(oracle):
    from tilecoderaytracer_amd import RtError, capi
    r = Renderer(_mixed_scene(HostScene.empty(), 3000))
    r.set_option("tables", 1)
    with pytest.raises(RtError) as e:
        r.render(8, 8, 2)
    assert e.value.code == capi.RT_ERR_CAPACITY
    r.set_option("tables", 0)
    assert r.render(8, 8, 2).shape == (8, 8, 3)


def test_counting_build_of_a_clustered_scene(oracle):
    """rt_render_stats on a scene with clustered runs: the counting variant of the clustered-scene kernel must render
    the oracle's image too, and count the same rays as the oracle traces."""
    want = oracle.OracleScene.grid(16, True).render(96, 80, 6)
    counters = oracle.OracleScene.counters()
    r = Renderer(HostScene.grid(16, True))
    img, st = r.render_stats(96, 80, 6)
    assert_same(img, want, "counting build")
    assert st["nearest_rays"] == counters.nearest_rays and st["shadow_rays"] == counters.shadow_rays
    assert st["nearest_scans_1_16"] + st["nearest_scans_17_32"] + st["nearest_scans_33_48"] + st["nearest_scans_49_64"] == st["wave_nearest_scans"]


def test_counting_build_keeps_tables_in_lds_whenever_they_fit(oracle):
    """rt_render_stats has no global-memory variant: with the automatic placement (tables = 0) it
    stages any scene that fits the 160 KiB of LDS -- two mirrors' 119 KB go to global memory in
    the product kernel, to LDS here -- and refuses only what cannot work: tables = 2."""
    from tilecoderaytracer_amd import RtError, capi
    want = oracle.OracleScene.two_mirrors().render(16, 16, 2)
    r = Renderer(HostScene.two_mirrors())                 # 119 KB of tables: global memory by default
    assert_same(r.render(16, 16, 2), want, "two mirrors, tables in global memory")
    assert r.launch_info().kernel == b"rt_render_kernel_large"
    img, _ = r.render_stats(16, 16, 2)
    assert_same(img, want, "counting build, two mirrors (automatic placement)")
    r.set_option("tables", 2)
    with pytest.raises(RtError) as e:
        r.render_stats(16, 16, 2)
    assert e.value.code == capi.RT_ERR_CAPACITY and "tables" in e.value.message
    r.set_option("tables", 1)                              # they do fit LDS
    img, _ = r.render_stats(16, 16, 2)
    assert_same(img, want, "counting build, two mirrors in LDS")


def test_launch_refuses_a_workgroup_beyond_the_kernels_launch_bounds(oracle):
    """block_threads up to 1024 is accepted as an option, but a launch whose kernel was compiled for fewer threads
    (256: the plain kernels; 512: the clustered-scene kernels and the counting builds) refuses it instead of faulting
    (round-2 experiment 18)."""
    from tilecoderaytracer_amd import RtError, capi
    for name, too_many in (("builtin", 512), ("builtin", 320), ("grid16", 576), ("grid16", 1024), ("twomirrors", 512)):
        r = Renderer(HostScene.named(name))
        r.set_option("block_threads", too_many)
        with pytest.raises(RtError) as e:
            r.render(64, 64, 2)
        assert e.value.code == capi.RT_ERR_INVALID and "launch bounds" in e.value.message
        r.set_option("block_threads", 256)
        assert_same(r.render(64, 64, 2), oracle.OracleScene.named(name).render(64, 64, 2), f"{name} after the refusal")
    r = Renderer(HostScene.named("grid32"))
    with pytest.raises(RtError):
        r.set_option("block_threads", 1088)
    with pytest.raises(RtError):
        r.set_option("block_threads", 100)


def test_fast_tables_against_item_tables(oracle):
    """Scenes without clustered runs: the kind-sorted item list with direct records (option fast = 1, the
    default) and the two item tables (fast = 0) render the same image, and so do the tight plane boxes."""
    want = oracle.OracleScene.builtin().render(200, 152, 6)
    r = Renderer(HostScene.builtin())
    assert_same(r.render(200, 152, 6), want, "fast tables")
    assert r.launch_info().kernel == b"rt_render_kernel"
    r.set_option("fast", 0)
    assert_same(r.render(200, 152, 6), want, "item tables")
    assert r.launch_info().kernel == b"rt_render_kernel_items"
    r.set_option("tight_planes", 0)
    assert_same(r.render(200, 152, 6), want, "item tables, sphere padding for planes")
    r.set_option("fast", 1)
    assert_same(r.render(200, 152, 6), want, "fast tables, sphere padding for planes")
