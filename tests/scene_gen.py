"""Seeded random scenes built through the reference-style verbs; works on both
HostScene (product host model) and OracleScene (oracle) because they expose
the same method names."""
import numpy as np

PALETTE = [(1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1), (1, 1, 1), (0.2, 0.2, 0.0), (0.33, 0.33, 0.33)]


def f32(x):
    return float(np.float32(x))


def build_random(scene, seed, n_spheres=12, n_finite=6, n_infinite=2, n_lights=2, shadows=True,
                 two_mirror_camera=True):
    """Populate `scene` (empty) with a random but sane arrangement in front of the
    setSceneTwoMirrors camera (eye at (0,-1,2.5) looking along +y)."""
    rng = np.random.RandomState(seed)

    def vec(lo, hi):
        return tuple(f32(v) for v in rng.uniform(lo, hi, 3))

    for k in range(n_lights):
        i = scene.add_sphere((f32(rng.uniform(-15, 15)), f32(rng.uniform(-5, 30)), f32(rng.uniform(6, 11))), f32(0.15))
        scene.set_light(i)
        scene.set_intensity(i, f32(rng.uniform(0.4, 1.0)))
    order = ["s"] * n_spheres + ["f"] * n_finite + ["i"] * n_infinite
    rng.shuffle(order)
    for kind in order:
        if kind == "s":
            i = scene.add_sphere((f32(rng.uniform(-8, 8)), f32(rng.uniform(4, 30)), f32(rng.uniform(0.3, 5))),
                                 f32(rng.uniform(0.3, 1.8)))
        elif kind == "f":
            o = (f32(rng.uniform(-8, 8)), f32(rng.uniform(5, 30)), f32(rng.uniform(0, 4)))
            if rng.rand() < 0.5:
                a = vec(-3, 3)
                b = vec(-3, 3)
                i = scene.add_finite_plane_corners(o, tuple(f32(o[j] + a[j]) for j in range(3)),
                                                   tuple(f32(o[j] + b[j]) for j in range(3)))
            else:
                i = scene.add_finite_plane_axes(o, vec(-1, 1), vec(-1, 1), f32(rng.uniform(1, 6)), f32(rng.uniform(1, 6)))
        else:
            up = rng.rand() < 0.5
            i = scene.add_infinite_plane((0.0, 0.0, 0.0 if up else 12.0), (0.0, f32(rng.uniform(-0.1, 0.1)), 1.0 if up else -1.0),
                                         (1.0, 0.0, 0.0))
        scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
        r = rng.rand()
        if r < 0.35:
            scene.set_reflective(i, f32(rng.choice([0.25, 0.5, 1.0])))
            scene.set_diffuse(i, f32(rng.choice([0.0, 0.5])))
        elif r < 0.7:
            scene.set_specular(i, f32(rng.uniform(0, 1)))
        if kind != "s" and rng.rand() < 0.5:
            scene.set_checkerboard(i, PALETTE[rng.randint(len(PALETTE))], PALETTE[rng.randint(len(PALETTE))],
                                   f32(rng.uniform(0.5, 4)), f32(rng.uniform(0.5, 4)))
    if shadows:
        scene.set_object_indices(0, 1)
    if two_mirror_camera:
        scene.camera_two_mirrors()
    return scene


def build_sphere_field(scene, seed, n_spheres=120, spread=60.0):
    """A run of consecutive spheres (long enough to be clustered by the kernel),
    of very different sizes and partly overlapping, over a reflective infinite
    ground and under a second, slightly tilted infinite plane, seen by the
    horizontal two-mirrors camera: the horizon rows hit the planes thousands of
    units away, from where shadow rays graze the whole field -- the regime in
    which the reference's float sphere test is coarse."""
    rng = np.random.RandomState(seed)
    for k in range(2):
        i = scene.add_sphere((f32(rng.uniform(-20, 20)), f32(rng.uniform(5, spread)), f32(rng.uniform(6, 11.5))), f32(0.15))
        scene.set_light(i)
        scene.set_intensity(i, f32(rng.uniform(0.5, 1.0)))
    g = scene.add_infinite_plane((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
    scene.set_reflective(g, 0.5)
    scene.set_diffuse(g, 0.5)
    scene.set_checkerboard(g, (1, 1, 1), (0, 0, 0), f32(rng.uniform(1, 5)), f32(rng.uniform(1, 5)))
    for k in range(n_spheres):
        r = f32(rng.choice([0.2, 0.5, 1.0, 1.0, 2.0, 3.5]))
        i = scene.add_sphere((f32(rng.uniform(-spread / 2, spread / 2)), f32(rng.uniform(4, spread)),
                              f32(rng.uniform(0.0, 4.0))), r)
        scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
        u = rng.rand()
        if u < 0.4:
            scene.set_reflective(i, f32(rng.choice([0.5, 1.0])))
            scene.set_diffuse(i, f32(rng.choice([0.0, 0.5])))
        elif u < 0.7:
            scene.set_specular(i, f32(rng.uniform(0, 1)))
    c = scene.add_infinite_plane((0.0, 0.0, 12.0), (0.0, f32(rng.uniform(-0.02, 0.02)), -1.0), (1.0, 0.0, 0.0))
    scene.set_reflective(c, 0.5)
    scene.set_specular(c, 0.5)
    scene.set_object_indices(0, 1)
    scene.camera_two_mirrors()
    return scene
