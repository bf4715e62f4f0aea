"""Randomised parity sweep, GPU kernel vs CPU oracle (development aid; the permanent cases live in tests/).

usage: fuzz_gpu.py [first_seed=1000] [count=200] [only=0..3] [far=1] [bigfields=1] [multi=1] [option=value ...]
multi=1: every scene through rt_render_multi with 2 ... 7 strips on this one device (TCRT_MULTI_ONE_DEVICE=1: measured cut, direct stores)
far=1: every scene is a far-origin grazing scene (scene_gen.build_far_grazing) in a strip 8 columns wide and 4 096 ... 32 768 rows
tall, whose rows around the middle hit the ground 1e4 ... 6e4 units away"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib                                   # noqa: E402
from scene_gen import build_far_grazing, build_random, build_room, build_sphere_field   # noqa: E402
from tilecoderaytracer_amd import HostScene, Renderer    # noqa: E402

kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
first, count = int(kv.get("first_seed", 1000)), int(kv.get("count", 200))
bad, t0 = [], time.time()
for seed in range(first, first + count):
    if "only" in kv and seed % 4 != int(kv["only"]):    # only=0: the clustered sphere fields, only=1: the rooms
        continue
    rng = np.random.RandomState(seed)
    if kv.get("far") == "1":
        mk = lambda s: build_far_grazing(s, seed)
        W, H, depth = 8, int(rng.choice([4096, 16384, 32768])), int(rng.randint(1, 5))
    elif seed % 4 == 1:                   # axis-aligned rooms: rectangles, slabs, lights hugging surfaces, scales (round 3's culls)
        mk = lambda s: build_room(s, seed)
        W, H, depth = int(rng.randint(8, 120)), int(rng.randint(8, 120)), int(rng.randint(0, 7))
    elif seed % 4 == 0:
        n = int(rng.choice([300, 600, 1000] if kv.get("bigfields") == "1" else [64, 80, 130, 260]))   # bigfields=1: 15-50 leaves (the voxel table's automatic range)
        mk = lambda s: build_sphere_field(s, seed, n_spheres=n, spread=float(rng.choice([30.0, 60.0, 200.0])))
        W, H, depth = 40, int(rng.choice([64, 512, 2048])), int(rng.randint(1, 6))
    else:
        kw = dict(n_spheres=int(rng.randint(0, 40)), n_finite=int(rng.randint(0, 12)), n_infinite=int(rng.randint(0, 3)),
                  n_lights=int(rng.randint(1, 4)), shadows=bool(rng.rand() < 0.8))
        mk = lambda s: build_random(s, seed, **kw)
        W, H, depth = int(rng.randint(1, 90)), int(rng.randint(1, 90)), int(rng.randint(0, 9))
    state = rng.get_state()
    host = mk(HostScene.empty())
    rng.set_state(state)
    orc = mk(oracle_lib.OracleScene())
    if kv.get("multi") in ("1", "2"):           # 2: LOOPBACK -- the strip-buffer transport and the trial too
        import ctypes as C
        from tilecoderaytracer_amd import capi
        os.environ["TCRT_MULTI_ONE_DEVICE"] = kv["multi"]
        ngpu = int(rng.choice([2, 3, 4, 5, 7]))
        got = np.full((W, H, 3), -3.0, np.float32)
        capi.check(capi.load_library().rt_render_multi(host.desc, host.camera, W, H, depth, ngpu, got.ctypes.data))
        want = orc.render(W, H, depth)
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            bad.append((seed, W, H, depth, len(d), d[0].tolist(), ngpu))
            print("MISMATCH", bad[-1], flush=True)
        if (seed - first) % 250 == 249:
            print(f"... {seed - first + 1} scenes, {len(bad)} mismatching, {time.time() - t0:.0f} s", flush=True)
        continue
    r = Renderer(host)
    for k, v in kv.items():                             # any other key=value: an rt_set_option for every scene (help=2 heavy=1 ...)
        if k not in ("first_seed", "count", "only", "learn", "far", "bigfields", "multi"):
            r.set_option(k, int(v))
    if seed % 3 == 0:
        r.set_option("tile_z", int(2 ** rng.randint(0, 7)))
    if kv.get("learn") == "1" and W > 0 and H > 0:      # learn=1: rt_learn_tile_order for the shape before it is rendered
        r.learn_tile_order(W, H, depth)
    got, want = r.render(W, H, depth), orc.render(W, H, depth)
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        bad.append((seed, W, H, depth, len(d), d[0].tolist()))
        print("MISMATCH", bad[-1], flush=True)
    if (seed - first) % 250 == 249:                  # a sign of life (a silent GPU job is taken for hung)
        print(f"... {seed - first + 1} scenes, {len(bad)} mismatching, {time.time() - t0:.0f} s", flush=True)
print(f"{count} scenes from seed {first}: {len(bad)} mismatching, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if bad else 0)
