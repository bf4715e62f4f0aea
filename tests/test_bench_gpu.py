"""bench.py's contract, checked on the GPU box: exactly one JSON line on stdout with
the fields the driver reads, the roofline and cpu_baseline objects included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--cpu-sample-columns", "64"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "Mrays/s" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 1000.0                                  # the north star's target was 100 Mrays/s
    assert abs(j["value"] - 4096 * 4096 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["algorithmic_bytes_per_launch"] == 4096 * 4096 * 12 and r["launches_per_frame"] == 1.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["parity"]["pixels_compared"] > 0 and c["parity"]["max_abs_delta"] == 0.0 and c["parity"]["pixels_over_1e-6"] == 0
    assert j["config"]["max_delta_vs_cpu_ref"] == 0.0
    assert j["sphere_grid"]["value"] > 100.0
    g = j["sphere_grid"]
    assert abs(g["hbm_write_roofline"]["frac"] - g["hbm_write_roofline"]["achieved"] / 8000.0) < 1e-5 and g["hbm_write_roofline"]["peak"] == 8000.0
    assert g["cpu_baseline"]["value"] > 0 and g["cpu_baseline"]["cores"] >= 1 and g["cpu_baseline"]["parity"]["max_abs_delta"] == 0.0
    assert g["cpu_baseline"]["parity"]["pixels_compared"] > 0
    sec = j["secondary"]
    assert sec["rays_per_pixel"] > 1.0 and sec["total_rays_per_s"] > j["value"] * 1e6
    assert 0.0 < sec["test_flop_frac_of_valu_peak"] < 1.0
    assert r["kernel"] == "rt_render_kernel"


@pytest.mark.gpu
def test_bench_force_dist_runs_the_chunked_single_frame_path_on_one_gpu():
    """--force-dist: torch.distributed over RCCL with one rank; --chunks 4: every frame in four column-chunk launches."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--force-dist", "--chunks", "4",
                          "--transport", "rccl", "--no-cpu-baseline", "--no-extra", "--workload", "grid16d8", "--size", "512"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert "4 column chunks" in j["config"]["partition"] and j["value"] > 10.0
    # what the chunks were gathered into is the frame one launch renders, bit for bit (bench.py measures that itself on a dist run)
    assert j["config"]["gathered_image_vs_one_gpu_frame"] == {"pixels_compared": 512 * 512, "pixels_differing": 0, "identical": True}
    assert j["config"]["transport"] == "rccl" and j["config"]["other_transport"] is None
    r = j["roofline"]
    assert r["kernel"].startswith("rt_render_kernel_clusters")
    # four launches per frame: a launch's share of the pixels over a launch's average duration, which is the frame's
    # pixels over the frame's kernel time -- not the whole strip over one chunk's time
    assert r["launches_per_frame"] == 4.0
    assert abs(r["algorithmic_bytes_per_launch"] - 512 * 512 * 12 / 4) < 1.0
    assert abs(r["frame_kernel_ms"] - 4 * r["kernel_ms"]) < 0.02 * r["frame_kernel_ms"] + 1e-3
    assert abs(r["achieved"] - 12 * 512 * 512 / (r["frame_kernel_ms"] * 1e-3) / 1e9) < 0.03 * r["achieved"] + 1e-3
    assert r["frame_kernel_ms"] <= j["ms_per_step"] * 1.05


@pytest.mark.gpu
def test_bench_shipped_workload_times_the_executable_end_to_end():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "shipped512", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["config"]["pixel_lines_md5"] == "ee680aed641062c6f3a5e0b3fba94199"       # the reference's 512x512 d3 text body
    assert j["config"]["identical_to_cpu_oracle_output"] is True
    assert j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["us_per_pixel"] > 0 and j["config"]["us_per_pixel"] > 0


SAME = {"pixels_compared": 512 * 512, "pixels_differing": 0, "identical": True}


@pytest.mark.gpu
def test_bench_force_dist_times_both_transports_on_one_gpu():
    """--force-dist with the default --transport auto: K steps through the strip buffer + RCCL, K steps with the kernel storing
    into the shared image; the faster is the headline, the other is beside it, and both images equal one GPU's frame."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--force-dist",
                          "--no-cpu-baseline", "--no-extra", "--workload", "grid16d8", "--size", "512"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    c = json.loads(lines[0])["config"]
    assert {c["transport"], c["other_transport"]["transport"]} == {"rccl", "direct"}
    assert c["gathered_image_vs_one_gpu_frame"] == SAME and c["other_transport"]["gathered_image_vs_one_gpu_frame"] == SAME
    assert c["other_transport"]["value"] > 10.0 and "transport_note" not in c


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_store_into_the_shared_image():
    """`bench.py --gpus 2` starts its own two ranks; TCRT_BENCH_ONE_DEVICE puts both on the box's one GPU and --backend gloo
    carries the control collectives (RCCL refuses two ranks on one device): the whole N > 1 sequence -- equal strips, every rank's
    kernel time, the re-cut, rt_learn_tile_order per strip, K timed frames -- with the direct transport,
    and rank 0's image compared with one GPU's frame."""
    env = dict(os.environ, TCRT_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
                          "--backend", "gloo", "--size", "512"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    c = j["config"]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 10.0
    assert c["transport"] == "direct" and c["other_transport"] is None
    assert c["gathered_image_vs_one_gpu_frame"] == SAME
    assert "2 x-strips of" in c["partition"] and "straight into rank 0's image" in c["partition"]
    assert "kernel ms per rank" in c["partition_note"] and "rt_learn_tile_order" in c["partition_note"]
    assert "MB" in c["scaling_note"] and "cpu_baseline" not in j
    assert j["pipelined"]["transport"] == "direct" and j["pipelined"]["value"] > 10.0 and "second shared image" in j["pipelined"]["what"]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_full_size_line():
    """The same at the default size with the sphere-grid line: its image, too, is compared with one GPU's frame."""
    env = dict(os.environ, TCRT_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "3",
                          "--backend", "gloo", "--no-pipelined"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    whole = {"pixels_compared": 4096 * 4096, "pixels_differing": 0, "identical": True}
    assert j["n_gpus"] == 2 and j["config"]["gathered_image_vs_one_gpu_frame"] == whole
    g = j["sphere_grid"]
    assert g["transport"] == "direct" and g["gathered_image_vs_one_gpu_frame"] == whole and g["value"] > 100.0
    assert g["hbm_write_roofline"]["peak"] == 16000.0 and "cpu_baseline" not in g


@pytest.mark.gpu
def test_bench_prefers_the_transport_whose_image_is_right():
    """Correctness before speed: with one pixel of the direct transport's image spoiled (TCRT_BENCH_CORRUPT, a testing aid) the
    headline is the strip-buffer transport whatever the times were, and the line says why."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--force-dist",
                          "--no-cpu-baseline", "--no-extra", "--workload", "grid16d8", "--size", "512"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, TCRT_BENCH_CORRUPT="direct"))
    assert out.returncode == 0, out.stderr[-2000:]
    c = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])["config"]
    assert c["transport"] == "rccl" and c["gathered_image_vs_one_gpu_frame"] == SAME
    other = c["other_transport"]
    assert other["transport"] == "direct" and other["gathered_image_vs_one_gpu_frame"]["pixels_differing"] == 1
    # (on one GPU the direct transport is the faster one -- no gather-to-self -- so the note must be there)
    if other["ms_per_step"] < json.loads([l for l in out.stdout.splitlines() if l.strip()][0])["ms_per_step"]:
        assert "differs from one GPU's frame" in c["transport_note"]
