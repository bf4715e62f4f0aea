"""raytracer_screen.txt: the host writer's fast exact "%f" formatter must emit
the same bytes as printf (the oracle's literal fprintf writer), and the text of
the built-in 512x512 depth-3 render must have the md5 SURVEY.md Appendix D
recorded for the reference's own output (pixel lines only)."""
import hashlib
import os

import numpy as np

from tilecoderaytracer_amd.host import write_screen_txt


def body_md5(path):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return hashlib.md5(b"\n".join(lines[10:])).hexdigest()      # tail -n +11


def test_writer_bytes_equal_printf(oracle, tmp_path):
    rng = np.random.RandomState(7)
    vals = np.concatenate([
        rng.uniform(0, 3, 3000), rng.uniform(-1e-3, 1e-3, 600), rng.uniform(-70000, 70000, 600),
        10.0 ** rng.uniform(-12, 12, 600), -(10.0 ** rng.uniform(-12, 12, 300)),
        [0.0, -0.0, 0.5e-6, 1.5e-6, 2.5e-6, 0.9999995, 0.99999949, 1.0, 0.75, 65535.0, 1e-45, -1e-45,
         3.4e38, -3.4e38, np.inf, -np.inf, np.nan, 0.0000005, 0.0000015, 123456.789, 2.6843546e8, 5.4e11, 5.6e11],
    ]).astype(np.float32)
    # exact ties at the 6th decimal that are representable in binary32
    ties = np.float32([k / 64.0 + 2.0 ** -21 * j for k in range(8) for j in range(4)])
    vals = np.concatenate([vals, ties])
    vals = np.resize(vals, (len(vals) // 3 + 1) * 3).reshape(-1, 1, 3)
    a, b = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    write_screen_txt(a, vals, 1.25, 3.5)
    oracle.write_screen_txt(b, vals, 1.25, 3.5)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_header_lines(tmp_path):
    p = str(tmp_path / "raytracer_screen.txt")
    write_screen_txt(p, np.zeros((2, 3, 3), np.float32), 0.5, 7.0)
    lines = open(p).read().split("\n")
    assert lines[:10] == [
        "OSX Awesome Picture", "Horizontal_Resolution:2.", "Vertical_Resolution:3.",
        "Hardware_Target:OSX C++.", "Number_of_Cores:1.", "IS_FOR_HARDWARE", "NO_PARTIONING",
        "Run_Time:0.500000.", "us/pixel:7.000000.", "filename:raytracer_screen.txt."]
    assert lines[10:16] == ["(0.000000, 0.000000, 0.000000)"] * 6 and lines[16] == ""


def test_header_names_the_gpus_and_the_partition(tmp_path):
    """--gpus G: CORE_NUM = G and the static partition's label (src/RayTracer.cpp:2037-2058)."""
    p = str(tmp_path / "raytracer_screen.txt")
    write_screen_txt(p, np.zeros((2, 3, 3), np.float32), 0.5, 7.0, n_cores=8)
    lines = open(p).read().split("\n")
    assert lines[4] == "Number_of_Cores:8." and lines[6] == "DUMB_STATIC_PARTIONING"
    assert lines[:4] == ["OSX Awesome Picture", "Horizontal_Resolution:2.", "Vertical_Resolution:3.", "Hardware_Target:OSX C++."]


def test_builtin_512_d3_text_md5_matches_the_reference(oracle, tmp_path):
    img = oracle.OracleScene.builtin().render(512, 512, 3)
    p = str(tmp_path / "raytracer_screen.txt")
    write_screen_txt(p, img, 0.4, 1.6)
    assert body_md5(p) == "ee680aed641062c6f3a5e0b3fba94199"
    assert os.path.getsize(p) > 512 * 512 * 31
