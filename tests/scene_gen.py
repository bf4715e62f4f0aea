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


def build_room(scene, seed):
    """Axis-aligned rooms: what round 3's culls changed.  Boxes of three-corner rectangles (the reference's makeSceneBox),
    nested and touching; axis-aligned INFINITE planes on any axis and side (slabs in the culls), lights that hug a surface
    (1e-3 ... 1e-1 in front of it) or sit exactly in a wall's plane, shading points that start on the surfaces (plane hits
    are offset 1e-3), a few spheres, everything scaled from 1e-2 to 1e3 and shifted away from the origin -- the culls' slack
    is relative to distance and magnitude -- seen by the horizontal two-mirrors camera (eye (0,-1,2.5), looking along +y)."""
    rng = np.random.RandomState(seed)
    scale = f32(rng.choice([0.01, 0.3, 1.0, 1.0, 7.0, 1000.0]))
    shift = np.float32(rng.choice([0.0, 0.0, 3.0, 250.0]) * rng.uniform(-1, 1, 3)) if scale <= 7.0 else np.zeros(3, np.float32)

    def pt(x, y, z):                    # a point of the unit-scale layout in front of the camera, scaled about the eye
        v = np.float32([x, y, z])
        eye = np.float32([0.0, -1.0, 2.5])
        w = eye + (v - eye) * np.float32(scale)
        return tuple(f32(c) for c in (w if scale > 7.0 else w + shift * 0))

    def box(o, dims, **mat):
        o, dims = np.float32(o), np.float32(dims)
        c = [o.copy() for _ in range(8)]
        c[1][0] += dims[0]; c[2][1] += dims[1]; c[3][2] += dims[2]
        c[4][0] += dims[0]; c[4][1] += dims[1]; c[5][0] += dims[0]; c[5][2] += dims[2]
        c[6][1] += dims[1]; c[6][2] += dims[2]; c[7] = o + dims
        for a, b, d in ((0, 3, 2), (0, 3, 1), (0, 1, 2), (7, 4, 6), (7, 4, 5), (7, 5, 6)):
            i = scene.add_finite_plane_corners(pt(*c[a]), pt(*c[b]), pt(*c[d]))
            scene.set_color(i, mat.get("color", (0.33, 0.33, 0.33)))
            if mat.get("reflective"):
                scene.set_reflective(i, mat["reflective"])
                scene.set_diffuse(i, mat.get("diffuse", 0.5))
            if mat.get("specular") is not None:
                scene.set_specular(i, mat["specular"])

    room_lo, room_hi = np.float32([-7, -3, -1]), np.float32([7, 14.5, 6])     # the eye is inside
    lights = []
    for k in range(int(rng.randint(1, 4))):
        mode = rng.randint(4)
        p = np.float32([rng.uniform(-6, 6), rng.uniform(2, 13), rng.uniform(0.5, 5)])
        if mode == 0:                   # hugging a wall of the room
            axis, side = rng.randint(3), rng.randint(2)
            gap = np.float32(rng.choice([1e-3, 1e-2, 1e-1]))
            p[axis] = room_hi[axis] - gap if side else room_lo[axis] + gap
        elif mode == 1:                 # exactly in a wall's plane (beside the wall's rectangle or on it)
            axis = rng.randint(3)
            p[axis] = room_hi[axis]
        lights.append(p)
        i = scene.add_sphere(pt(*p), f32(0.15 * scale))
        scene.set_light(i)
        scene.set_intensity(i, f32(rng.uniform(0.4, 1.0)))
    order = ["room", "slab", "table", "inner"] + ["sphere"] * int(rng.randint(0, 7)) + ["plane"] * int(rng.randint(0, 4))
    rng.shuffle(order)
    for what in order:
        if what == "room":
            box(room_lo, room_hi - room_lo, color=(0.33, 0.33, 0.33), specular=0.0)
        elif what == "slab":            # a ceiling slab with a gap to the walls, like the reference's scene
            box((-6, 1.5, 5), (12, 12, 1), color=(0.66, 0.66, 0.66), reflective=0.5, specular=0.5)
        elif what == "table":
            o = (f32(rng.uniform(-4, 2)), f32(rng.uniform(3, 9)), 0.0)
            box(o, (f32(rng.uniform(0.5, 3)), f32(rng.uniform(0.5, 3)), f32(rng.uniform(0.2, 2))), color=(0.2, 0.2, 0.0),
                reflective=float(rng.choice([0.0, 0.5])), specular=0.2)
        elif what == "inner":           # a box that touches the floor plane z = 0 and another box's face (coincident surfaces)
            box((-0.5, 5.0, 0.0), (1, 1, 1), color=(0.2, 0.2, 0.0))
            box((0.5, 5.0, 0.0), (0.7, 1, 0.5), color=(1, 0, 0), reflective=1.0, diffuse=0.0)
        elif what == "sphere":
            i = scene.add_sphere(pt(rng.uniform(-5, 5), rng.uniform(2, 12), rng.uniform(0.3, 4)), f32(rng.uniform(0.2, 1.2) * scale))
            scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
            if rng.rand() < 0.5:
                scene.set_reflective(i, f32(rng.choice([0.5, 1.0])))
                scene.set_diffuse(i, f32(rng.choice([0.0, 0.5])))
        else:                           # an axis-aligned infinite plane: any axis, either side, through or beside the room
            axis, sign = rng.randint(3), float(rng.choice([-1.0, 1.0]))
            n = [0.0, 0.0, 0.0]; n[axis] = sign
            h = [0.0, 0.0, 0.0]; h[(axis + 1) % 3] = 1.0
            where = np.float32([0, 5, 0])
            where[axis] = np.float32(rng.choice([-20.0, 40.0, 13.0] if axis == 1 else [0.0, -1.0, 6.0, 3.3, -20.0, 40.0]))   # never across the eye's view
            i = scene.add_infinite_plane(pt(*where), tuple(n), tuple(h))
            scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
            if rng.rand() < 0.6:
                scene.set_reflective(i, 0.5)
                scene.set_diffuse(i, 0.5)
            if rng.rand() < 0.5:
                scene.set_checkerboard(i, (1, 1, 1), (0, 0, 0), f32(3 * scale), f32(3 * scale))
    if rng.rand() < 0.85:
        scene.set_object_indices(0, 1)
    scene.camera_two_mirrors()
    return scene


def build_far_grazing(scene, seed):
    """Rays that ARRIVE from far away at tilted rectangles near the origin, grazing them (round-3 advisor's case for the tight
    plane boxes, csrc/rt_capi.hip box_item()): the hit parameter t = num / den of a ray that starts 1e4-6e4 units away loses
    about 1e-6 of that distance to cancellation in n.o + dto, and t d_k + o_k cancels again.
      * shadow rays: the horizontal two-mirrors camera (eye (0,-1,2.5) looking along +y) sees a ground plane to the horizon
        -- in a strip thousands of rows tall the rows just below the middle hit it 1e4 ... 6e4 units away -- and the lights sit
        among the rectangles, near the origin, so those shadow segments come in almost level and graze the nearly level plates;
      * nearest-hit rays: a mirror wall y = D (D = 1e4 ... 6e4) sends the camera rays back the same way.
    Plates: nearly horizontal (normal (a, b, 1), |a|, |b| <= 0.05), nearly vertical facing the rays, and arbitrary; some are
    mirrors.  Same calls on HostScene and OracleScene."""
    rng = np.random.RandomState(seed)
    for k in range(int(rng.randint(1, 3))):
        i = scene.add_sphere((f32(rng.uniform(-3, 3)), f32(rng.uniform(3, 9)), f32(rng.uniform(1.0, 5.0))), f32(0.15))
        scene.set_light(i)
        scene.set_intensity(i, f32(rng.uniform(0.5, 1.0)))
    i = scene.add_infinite_plane((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
    scene.set_color(i, (0, 1, 0))
    scene.set_reflective(i, f32(rng.choice([0.0, 0.5])))
    scene.set_diffuse(i, 0.5)
    if rng.rand() < 0.5:
        scene.set_checkerboard(i, (1, 1, 1), (0, 0, 0), 3.0, 3.0)
    far = f32(rng.choice([1.0e4, 3.0e4, 6.0e4]))
    i = scene.add_infinite_plane((0.0, far, 0.0), (0.0, -1.0, 0.0), (1.0, 0.0, 0.0))
    scene.set_color(i, (1, 1, 1))
    scene.set_reflective(i, 1.0)
    scene.set_diffuse(i, 0.0)
    for k in range(int(rng.randint(4, 14))):
        o = (f32(rng.uniform(-5, 3)), f32(rng.uniform(2, 14)), f32(rng.uniform(0.2, 4.5)))
        style = rng.randint(3)
        if style == 0:          # nearly level plate: level shadow rays graze it
            n = (f32(rng.uniform(-0.05, 0.05)), f32(rng.uniform(-0.05, 0.05)), f32(rng.choice([-1.0, 1.0])))
            h = (1.0, f32(rng.uniform(-0.3, 0.3)), 0.0)
        elif style == 1:        # nearly vertical, along the rays' way: grazed by rays along +-y
            n = (f32(rng.choice([-1.0, 1.0])), f32(rng.uniform(-0.03, 0.03)), f32(rng.uniform(-0.03, 0.03)))
            h = (0.0, 1.0, f32(rng.uniform(-0.2, 0.2)))
        else:
            n = tuple(f32(v) for v in rng.uniform(-1, 1, 3))
            h = tuple(f32(v) for v in rng.uniform(-1, 1, 3))
        i = scene.add_finite_plane_axes(o, n, h, f32(rng.uniform(1, 8)), f32(rng.uniform(1, 8)))
        scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
        r = rng.rand()
        if r < 0.3:
            scene.set_reflective(i, f32(rng.choice([0.5, 1.0])))
            scene.set_diffuse(i, f32(rng.choice([0.0, 0.5])))
        elif r < 0.6:
            scene.set_specular(i, f32(rng.uniform(0, 1)))
    for k in range(int(rng.randint(0, 4))):
        i = scene.add_sphere((f32(rng.uniform(-4, 4)), f32(rng.uniform(3, 12)), f32(rng.uniform(0.5, 3))), f32(rng.uniform(0.3, 1.2)))
        scene.set_color(i, PALETTE[rng.randint(len(PALETTE))])
    scene.set_object_indices(0, 1)
    scene.camera_two_mirrors()
    return scene
