#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one full render of the workload image.  At N = 1 the whole image
is one kernel launch on one GPU.  At N > 1 there is one process per GPU: started
by the driver with torch.distributed.run, or -- `python bench.py --gpus N` with no
rank variables in the environment -- by this file itself, which starts
`python -m torch.distributed.run ... bench.py --gpus N ...` as a child process
before it has imported torch or touched HIP, relays rank 0's JSON line and exits
with the child's code (self_launch(); the reference's main() spawns its own ranks
too, src/RayTracer.cpp:1536-1566).  The image is split into N contiguous x-strips
(the framebuffer is x-major, pixels[x][z], so a strip is one contiguous
block), every rank renders its strip into HBM and the strips travel to rank 0
over RCCL (torch.distributed backend "nccl") -- STRONG scaling: the image is
fixed, the work per GPU shrinks.  The headline `value` is the SINGLE-FRAME
figure SURVEY.md 8(d) defines: every frame is rendered and delivered before
the next one starts.  Within a frame a strip may be rendered and sent in
column CHUNKS (--chunks; automatic: from the measured kernel and transfer
times), chunk k on its way while chunk k+1 is rendered, as the reference's
ranks write into the shared image while they render.  The strips are cut by
measured cost (--partition balanced, the default): the first two warm-up steps
(two extra untimed ones if W < 3) run N equal strips and measure every rank's
kernel time and the time of a gather on its own; the image is then re-cut so
that rank 0 -- which receives and sends nothing -- renders as long as a peer
needs to render and ship its columns (tilecoderaytracer_amd/distributed.py:
balanced_bounds).  When a strip is one launch, each rank then lets the library
measure where that launch shape should start handing out its tile rows
(rt_learn_tile_order: one counting frame and twenty timed ones, before the
timed region; --no-learn skips it; scheduling only, same pixels).  All W
warm-up steps are untimed; the K timed steps all run the final partition.  The throughput of a STREAM of frames (gather of frame k
under the render of frame k+1) is measured afterwards and reported beside the
headline as `pipelined`, labelled as a different figure.

TRANSPORTS (--transport, N > 1).  The strips can reach rank 0 in two ways, and which is faster is a property of the
machine: "rccl" = strip buffers sent with RCCL as described above; "direct" = rank 0's image is shared with the other
ranks over HIP IPC (rt_shared_image_*, include/rt_capi.h) and every rank's kernel stores its strip straight into it --
the reference's ranks writing into the one `pixels` array (src/RayTracer.cpp:904-923, 1188-1196) -- with a one-word
all-reduce in stream order behind the kernel as the frame's fence; strips of equal measured kernel time.  The default,
"auto", times K steps of EACH (own warm-up, own partition), reports the faster as the headline (`config.transport`) and
the other beside it (`config.other_transport`), and compares both images, bit for bit, with the frame rank 0's GPU
renders alone.  --backend gloo with TCRT_BENCH_ONE_DEVICE=1 is a rehearsal aid: several ranks on ONE GPU (RCCL refuses
that), direct transport only -- the whole N > 1 sequence on a one-GPU box (tests/test_bench_gpu.py).

Timed region: barrier + synchronize, K steps (kernel + gather), barrier +
synchronize; MAX over ranks.  The framebuffer stays in HBM (inputs -- the
scene tables -- are resident before the region starts); no device-to-host copy
is inside it.  value = W*H*K / t  [Mrays/s], primary rays = pixels, the
reference's own figure of merit (src/RayTracer.cpp:1104, us/pixel inverted).

Extra objects on the JSON line:
  roofline      the dominant (only) kernel -- its name as launched is in
                roofline.kernel -- against the HBM roofline the north star
                names: achieved = algorithmic bytes (12 B per pixel: one packed
                fp32 RGB store) / average kernel duration, measured with HIP
                events on the launch stream inside the timed region.
  sphere_grid   (default workload only) the same partition timed on the
                1024-sphere grid scene, BASELINE.json configs[2], the scene the
                north star quotes for the 1/2/4/8-GPU series; a few steps, after
                and outside the headline timed region.
  secondary     SURVEY.md 8(d)'s secondary figures from the counting build: all rays
                per second (primary + reflection + shadow) and intersection tests
                per second against the plain fp32 VALU peak.
  cpu_baseline  the CPU oracle (oracle/rt_oracle.c, a port of the reference's
                algorithm; the reference itself is unbuildable here) timed on
                this host's cores on a bounded sample of the same workload.

--workload shipped / shipped512 time the drop-in EXECUTABLE end to end instead
(process start to raytracer_screen.txt on disk: the reference's own Run_Time /
us/pixel, src/RayTracer.cpp:1089-1104, 1576-1577) beside the CPU oracle doing
the same job; see shipped_workload().
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# workload name -> (scene, W, H, max_depth, BASELINE.json config it corresponds to)
WORKLOADS = {
    "builtin":          ("builtin",         4096, 4096, 4, "configs[1]: built-in Scene, 4096x4096, depth 4"),
    "grid32":           ("grid32",          4096, 4096, 4, "configs[2]: 1024-sphere grid + 2 planes, 4096x4096, depth 4 (shadow scan on)"),
    "grid32-noshadow":  ("grid32-noshadow", 4096, 4096, 4, "configs[2] variant: shadow scan range [0,0) as in the survey fixtures"),
    "grid16d8":         ("grid16",          4096, 4096, 8, "configs[4]: 256-sphere grid, 4096x4096, depth 8 (shadow scan on)"),
    "builtin8k":        ("builtin",         8192, 8192, 4, "configs[3]: 8192x8192 tiled one strip per GPU"),
    "twomirrors":       ("twomirrors",      4096, 4096, 50, "not a BASELINE config: the reference's SCENE 2 (3 920 objects, facing mirrors, "
                                                             "src/Scene.cpp:23-206), MAX_RECURSION_LEVEL 50; tables of 119 KB read from global memory"),
    # the whole drop-in program, process start -> raytracer_screen.txt (shipped_workload())
    "shipped":          ("builtin",         500, 504, 50, "the reference as shipped: SCREEN 500x504, MAX_RECURSION_LEVEL 50 (src/rt_project_parameters.h:65-66,73)"),
    "shipped512":       ("builtin",         512, 512, 3, "configs[0]: built-in Scene, 512x512, depth 3, .txt output"),
}

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL = 12             # 3 x fp32 framebuffer store, SURVEY.md section 8(d)
CLOCK_HZ = 2.4e9                 # MI355X_MICROARCH.md: 2.4 GHz peak engine clock
N_SIMDS = 256 * 4                # 256 CUs x 4 SIMD-32
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * CLOCK_HZ     # one non-fused fp32 op per SIMD lane and clock (157.3 TF counts FMA = 2)
KERNEL_SOURCES = ("rt_kernel.hip", "rt_capi.hip", "rt_tables.h")


def kernel_source_digest():
    """SHA-256 over the kernel's source files: ties a counter profile to the build it was taken from."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "tilecoderaytracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


# What a CU of gfx950 issues per cycle (scripts/ubench/issue_rate.hip, profiles/r04_issue_rate_ubench.txt, at the nominal 2.4 GHz):
# 1.75 wave64 vector instructions (four SIMDs), and ONE scalar-unit instruction (0.97 measured: the scalar unit is shared by the
# CU's four SIMDs; branches go through it too).  A 2 : 1 mix of vector and scalar instructions tops out at 1.56 + 0.78.
VECTOR_PER_CU_CYCLE = 1.75
SCALAR_PER_CU_CYCLE = 0.97
N_CUS = N_SIMDS // 4


def valu_issue_roofline(rec, kernel_ms):
    """The binding roofline of these kernels: instruction issue.  SQ_INSTS_VALU and SQ_INSTS_SALU (+ SQ_INSTS_BRANCH where the
    profile has it) per launch, per CU and cycle of the kernel (duration x 2.4 GHz), against what a CU was measured to issue.
    `frac` is the vector figure; the scalar unit's is beside it -- the two limits are reached together by a 2 : 1 mix at 0.89 and
    0.80 of them.  (The fractions of rounds 2-3 -- one vector instruction per 2 / 2.5 cycles and SIMD -- are kept for comparison.)"""
    insts = float(rec["sq_insts_valu"])
    scalar = float(rec.get("sq_insts_salu") or 0.0) + float(rec.get("sq_insts_branch") or 0.0)
    kernel_cycles = kernel_ms * 1e-3 * CLOCK_HZ
    at2 = insts / N_SIMDS * 2.0
    at25 = insts / N_SIMDS * 2.5
    per_cu = (lambda n: n / N_CUS / kernel_cycles) if kernel_cycles > 0 else (lambda n: None)
    vec, sca = per_cu(insts), per_cu(scalar)
    return {
        "bound": "instruction_issue",
        "sq_insts_valu_per_launch": insts,
        "sq_insts_salu_per_launch": rec.get("sq_insts_salu"),
        "sq_insts_branch_per_launch": rec.get("sq_insts_branch"),
        "sq_busy_cycles_per_engine": rec.get("sq_busy_cycles_per_engine"),
        "kernel_cycles_at_2.4GHz": round(kernel_cycles, 0),
        "vector_per_cu_and_cycle": round(vec, 3) if vec is not None else None,
        "scalar_and_branch_per_cu_and_cycle": round(sca, 3) if sca is not None else None,
        "peak_vector_per_cu_and_cycle": VECTOR_PER_CU_CYCLE,
        "peak_scalar_per_cu_and_cycle": SCALAR_PER_CU_CYCLE,
        "frac": round(vec / VECTOR_PER_CU_CYCLE, 4) if vec is not None else None,
        "frac_scalar_unit": round(sca / SCALAR_PER_CU_CYCLE, 4) if sca is not None else None,
        "valu_issue_cycles_per_simd_at_2": round(at2, 0),
        "frac_at_2_cycles_per_wave64_op": round(at2 / kernel_cycles, 4) if kernel_cycles > 0 else None,
        "frac_at_2.5_cycles_measured": round(at25 / kernel_cycles, 4) if kernel_cycles > 0 else None,
        "source": "SQ_INSTS_* from profiles/pmc_traffic.json (profiles/*_pmc_sq.csv), kernel time from this run's HIP events; "
                  "peaks from scripts/ubench/issue_rate.hip (profiles/r04_issue_rate_ubench.txt)",
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="builtin", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=0, help="override W=H (a non-standard run; recorded in config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short sphere-grid measurement that follows the default workload")
    ap.add_argument("--cpu-sample-columns", type=int, default=0,
                    help="columns of the image the CPU oracle renders (default: sized for ~10-30 CPU-seconds)")
    ap.add_argument("--tile-z", type=int, default=0, help="wavefront tile height override (speed only)")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="rt_set_option tuning knob (speed only), e.g. --option stack=2")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the gather even at N = 1 (exercises the RCCL path on one GPU)")
    ap.add_argument("--partition", default="balanced", choices=["balanced", "equal"],
                    help="N > 1: 'balanced' re-cuts the strips from kernel and gather times measured in the first "
                         "warm-up steps (rank 0, which receives, renders more when a link is slower than a GPU); "
                         "'equal' keeps N equal strips")
    ap.add_argument("--no-overlap", action="store_true",
                    help="(accepted for compatibility; the headline at N > 1 is always the single-frame figure)")
    ap.add_argument("--no-learn", action="store_true", help="N > 1: do not call rt_learn_tile_order for the ranks' strips")
    ap.add_argument("--chunks", type=int, default=0,
                    help="N > 1, single-frame mode: column chunks a strip is rendered and sent in (chunk k travels while "
                         "chunk k+1 is rendered); 0 = automatic from the measured kernel and transfer times, 1 = none")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "direct"],
                    help="N > 1: how the strips reach rank 0.  'rccl': strip buffers + RCCL point-to-point transfers (in column chunks); "
                         "'direct': rank 0's image is shared over HIP IPC and every rank's kernel stores its strip straight into it "
                         "(the reference's shared `pixels`); 'auto' times K steps of each and reports the faster as the headline, "
                         "the other beside it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the control collectives; gloo is a testing aid (several ranks on one GPU "
                         "with TCRT_BENCH_ONE_DEVICE=1: RCCL refuses that) and carries the direct transport only")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="N > 1: skip the second, pipelined measurement (gather of frame k under the render of frame k+1)")
    return ap.parse_args(argv)


def host_cores():
    """(cpus, note): one logical CPU per PHYSICAL core this process may use -- the affinity
    mask, one SMT sibling per (package, core) of /proc/cpuinfo -- capped by the cgroup CPU
    quota (a container can see 256 CPUs and be allowed 16).  The CPU leg pins one thread to each."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    core_of = {}
    try:
        cpu = None
        phys = core = 0
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, phys, core = int(v), 0, None
            elif k == "physical id":
                phys = int(v)
            elif k == "core id":
                core = int(v)
            elif not k and cpu is not None:
                core_of[cpu] = (phys, core if core is not None else cpu)
                cpu = None
    except Exception:
        core_of = {}
    seen, cpus = set(), []
    for c in allowed:
        key = core_of.get(c, ("?", c))
        if key not in seen:
            seen.add(key)
            cpus.append(c)
    physical = len(cpus)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = max(1, int(float(txt[0]) / float(txt[1])))
            else:
                q = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    quota = max(1, q // period)
            break
        except Exception:
            continue
    n = min(physical, quota or physical, 64)
    if n < physical:
        # fewer threads than cores (a quota inside a big machine shared with other jobs): leave the
        # placement to the scheduler, which avoids busy cores; pinning could land on one
        pin = False
        cpus = cpus[:n]
    else:
        pin = True
    note = (f"{len(allowed)} logical CPUs in the affinity mask = {physical} physical cores"
            + (f", cgroup quota {quota}" if quota else "") + f"; {len(cpus)} threads"
            + (", one pinned per physical core" if pin else ", placed by the scheduler"))
    if not pin:
        cpus = [-1] * len(cpus)
    return cpus, note


def cpu_baseline(scene_name, W, H, depth, sample_columns, gpu_image=None, repeats=3):
    """Time the CPU oracle on the host: one core (the reference's shipped mode,
    PARTIONING_STRATEGY 0) and all physical cores with the reference's static partitioning
    (strategy 1, src/RayTracer.cpp:904-923 with CORE_NUM > 1: contiguous shares, C threads inside
    the oracle harness, oracle/rt_oracle_mt.c), median of `repeats` runs.  With gpu_image (the
    GPU's W x H x 3 frame on the host) every sampled column is also compared with it -- the
    "max per-channel delta vs CPU ref" half of the metric -- after the clocks have stopped."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib  # noqa: E402  (the oracle is the thing timed here, nothing else uses it)

    cpus, core_note = host_cores()
    cores = len(cpus)
    chunk = 8
    n_chunks = min(max(cores, sample_columns // chunk), W // chunk)
    starts = [int(i * (W - chunk) / max(n_chunks - 1, 1)) for i in range(n_chunks)]      # ascending, evenly spread
    scene = oracle_lib.OracleScene.named(scene_name)
    out = np.empty((n_chunks * chunk, H, 3), dtype=np.float32)
    c_starts = (C.c_int * n_chunks)(*starts)
    c_cpus = (C.c_int * cores)(*cpus)
    pixels = n_chunks * chunk * H

    def run(threads):
        sec = C.c_double()
        rc = oracle_lib.LIB.orc_render_static_partition(scene.h, C.byref(scene.cam), W, H, depth, c_starts, n_chunks, chunk,
                                                        threads, c_cpus, out.ctypes.data, C.byref(sec))
        if rc:
            raise RuntimeError("orc_render_static_partition failed")
        return sec.value

    # one core: a sixteenth of the sample is enough for a steady figure
    one_chunks = max(1, n_chunks // 16)
    one_starts = (C.c_int * one_chunks)(*starts[:: max(1, n_chunks // one_chunks)][:one_chunks])
    one_times = []
    for _ in range(repeats):
        sec = C.c_double()
        oracle_lib.LIB.orc_render_static_partition(scene.h, C.byref(scene.cam), W, H, depth, one_starts, one_chunks, chunk, 1,
                                                   c_cpus, out.ctypes.data, C.byref(sec))
        one_times.append(sec.value)
    one_core = one_chunks * chunk * H / float(np.median(one_times)) / 1e6
    times = [run(cores) for _ in range(repeats)]
    dt = float(np.median(times))
    parity = None
    if gpu_image is not None:
        worst, over6, over4 = 0.0, 0, 0
        for k, x0 in enumerate(starts):
            d = np.abs(gpu_image[x0:x0 + chunk].astype(np.float64) - out[k * chunk:(k + 1) * chunk].astype(np.float64)).max(axis=-1)
            d = np.where(np.isnan(d), np.inf, d)
            worst = max(worst, float(d.max()))
            over6 += int((d > 1e-6).sum())
            over4 += int((d > 1e-4).sum())
        parity = {"pixels_compared": pixels, "max_abs_delta": worst, "pixels_over_1e-6": over6, "pixels_over_1e-4": over4}
    return {
        "value": round(pixels / dt / 1e6, 4),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_chunks} chunks of {chunk} columns x {H} rows spread over the {W}x{H} image ({pixels} pixels), "
                  f"dealt to the threads in contiguous shares (the reference's static partitioning); median of {repeats} runs, "
                  f"{dt:.2f} s wall each (all: {[round(t, 2) for t in times]}); CPU oracle oracle/rt_oracle.c, gcc -O2 -ffp-contract=off; "
                  + core_note,
        "one_core_value": round(one_core, 4),
        "parity": parity,
    }


def shipped_workload(args):
    """The reference's own run: main() renders SCREEN_WIDTH x SCREEN_HEIGHT at MAX_RECURSION_LEVEL and
    writes raytracer_screen.txt; its figure of merit is Run_Time and us/pixel of the whole program
    (src/RayTracer.cpp:1089-1104, 1576-1577).  Here: bin/tcrt_raytracer (the drop-in executable: scene,
    HIP context, render, device-to-host copy, text formatting, file on disk) timed as a process, and
    the CPU oracle doing the same job in this process (render on one core -- the reference's shipped
    CORE_NUM 1 -- then the same byte-exact writer), both `steps` times, medians.  One JSON line."""
    import hashlib
    import subprocess
    import tempfile
    scene_name, W, H, depth, cfg_note = WORKLOADS[args.workload]
    exe = os.path.join(ROOT, "tilecoderaytracer_amd", "bin", "tcrt_raytracer")
    if not os.path.exists(exe):
        sys.exit(f"{exe} not built (make -C tilecoderaytracer_amd/csrc)")
    steps = max(1, min(args.steps, 10))
    gpu_wall, gpu_md5 = [], None
    with tempfile.TemporaryDirectory() as d:
        out_txt = os.path.join(d, "raytracer_screen.txt")
        for k in range(args.warmup and 1 or 0):
            subprocess.run([exe, "--width", str(W), "--height", str(H), "--depth", str(depth), "--out", out_txt],
                           check=True, capture_output=True)
        for k in range(steps):
            t0 = time.perf_counter()
            subprocess.run([exe, "--width", str(W), "--height", str(H), "--depth", str(depth), "--out", out_txt],
                           check=True, capture_output=True)
            gpu_wall.append(time.perf_counter() - t0)
        body = b"".join(open(out_txt, "rb").readlines()[10:])           # the pixel lines (the header carries the timing)
        gpu_md5 = hashlib.md5(body).hexdigest()
        gpu_bytes = os.path.getsize(out_txt)
        # the CPU oracle doing the same job: render (one core), then the same writer
        cpu = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib  # noqa: E402  (the checker, timed beside the product)
            from tilecoderaytracer_amd import host as host_mod
            scene = oracle_lib.OracleScene.named(scene_name)
            cpu_wall = []
            cpu_txt = os.path.join(d, "cpu_screen.txt")
            for k in range(min(steps, 3)):
                t0 = time.perf_counter()
                img = scene.render(W, H, depth)
                t1 = time.perf_counter()
                host_mod.write_screen_txt(cpu_txt, img, run_time_s=t1 - t0)
                cpu_wall.append((time.perf_counter() - t0, t1 - t0))
            cpu_body = b"".join(open(cpu_txt, "rb").readlines()[10:])
            cpu = {"wall_s": round(float(np.median([w for w, _ in cpu_wall])), 4),
                   "render_s": round(float(np.median([r for _, r in cpu_wall])), 4),
                   "us_per_pixel": round(float(np.median([w for w, _ in cpu_wall])) / (W * H) * 1e6, 4),
                   "cores": 1, "kind": "port",
                   "pixel_lines_md5": hashlib.md5(cpu_body).hexdigest(),
                   "what": "oracle/rt_oracle.c on one core (the reference's shipped CORE_NUM 1) + the same .txt writer, in this process"}
    wall = float(np.median(gpu_wall))
    out = {
        "metric": "Mrays/sec of the whole drop-in program (process start to raytracer_screen.txt on disk); the reference's Run_Time / us/pixel",
        "value": round(W * H / wall / 1e6, 4),
        "unit": "Mrays/s",
        "n_gpus": 1, "steps": steps, "warmup": 1 if args.warmup else 0,
        "ms_per_step": round(wall * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (hard-coded reference scene; no files)",
        "config": {"workload": f"bin/tcrt_raytracer, {scene_name} scene, {W}x{H}, max depth {depth}, .txt output ({gpu_bytes} bytes)",
                   "baseline_config": cfg_note,
                   "us_per_pixel": round(wall / (W * H) * 1e6, 4),
                   "runs_s": [round(w, 4) for w in gpu_wall],
                   "pixel_lines_md5": gpu_md5,
                   "identical_to_cpu_oracle_output": (cpu is not None and cpu["pixel_lines_md5"] == gpu_md5) if cpu else None,
                   "note": "dominated by process start, HIP context creation and the first kernel load; the kernel itself is "
                           "microseconds at this size (the N = 1 default workload is the kernel figure)"},
        "roofline": None,
        "cpu_baseline": ({"value": round(W * H / cpu["wall_s"] / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                          "sample": f"the whole {W}x{H} job, median of {min(steps, 3)} runs", **cpu} if cpu else None),
    }
    print(json.dumps(out), flush=True)


def free_port():
    """A TCP port nobody listens on right now (the rendezvous of the ranks this process starts)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(gpus, argv, port):
    """argv of the child that runs `gpus` ranks of this file, one per GPU, over RCCL: what the reference's main() does
    with ilib_proc_exec (src/RayTracer.cpp:1536-1566: the first process spawns the others and they meet at a barrier).
    TCRT_BENCH_LAUNCHER (tests) replaces `python -m torch.distributed.run` by another program with the same arguments."""
    launcher = os.environ.get("TCRT_BENCH_LAUNCHER")
    head = launcher.split() if launcher else [sys.executable, "-m", "torch.distributed.run"]
    return head + ["--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                   os.path.abspath(__file__)] + list(argv)


def launch_environment(environ=None):
    """The children's environment: marked as started from here (never a second generation), dmabuf IPC as this pool's
    driver needs for RCCL between processes, and no rank variables inherited from whoever started this process."""
    env = dict(os.environ if environ is None else environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["TCRT_BENCH_CHILD"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def result_lines(text):
    """The bench lines in a child's stdout: JSON objects that carry "metric" (anything else a launcher or a library
    printed there is noise and goes to stderr)."""
    found, noise = [], []
    for line in text.splitlines():
        s = line.strip()
        ok = False
        if s.startswith("{") and s.endswith("}"):
            try:
                ok = "metric" in json.loads(s)
            except ValueError:
                ok = False
        (found if ok else noise).append(line)
    return found, noise


def self_launch(args, argv):
    """`python bench.py --gpus N` with WORLD_SIZE unset: start the N ranks as a CHILD process -- before this process has
    imported torch or touched HIP, and never by exec (a process that has initialised the GPU must not be replaced) --,
    relay rank 0's single JSON line on stdout, everything else on stderr, and return the child's exit code."""
    import subprocess
    assert "torch" not in sys.modules, "the launcher must not have initialised torch / HIP"
    cmd = launch_command(args.gpus, argv, free_port())
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, env=launch_environment(), text=True)
    found, noise = result_lines(child.stdout or "")
    for line in noise:
        print(line, file=sys.stderr)
    sys.stderr.flush()
    if child.returncode != 0:
        print(f"bench.py: the ranks failed (exit code {child.returncode})", file=sys.stderr, flush=True)
        return child.returncode
    if len(found) != 1:
        print(f"bench.py: expected one result line from rank 0, got {len(found)}", file=sys.stderr, flush=True)
        return 1
    print(found[0], flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.workload in ("shipped", "shipped512"):
        if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
            sys.exit("the shipped workloads time one process on one GPU")
        return shipped_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ and not os.environ.get("TCRT_BENCH_CHILD"):
            sys.exit(self_launch(args, argv))          # one process per GPU: started from here
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    # (this pool's driver shares device memory between processes -- RCCL's buffers, rt_shared_image_* -- through dmabuf only;
    # the variable is read when the HIP runtime starts, i.e. below)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from tilecoderaytracer_amd import HostScene, Renderer
    from tilecoderaytracer_amd.distributed import DirectStrips, SharedImage, StripPipeline, balance_direct, measure_and_balance

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path is HIP-only (no CPU fallback)")
    if os.environ.get("TCRT_BENCH_ONE_DEVICE"):        # testing aid: every rank on device 0 (if the RCCL build allows it)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # RCCL prints its version banner on stdout when NCCL_DEBUG is set (it is, on this pool);
        # stdout must carry exactly one JSON line, so route fd 1 to stderr until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where the control collectives' small tensors live
    shared_images = []

    stream = torch.cuda.current_stream(dev).cuda_stream

    def measure(workload, steps, warmup, size=0, overlap=False, transport="rccl"):
        """Warm up, then time exactly `steps` steps of `workload` between two fences
        (barrier + synchronize); returns the MAX over ranks.  overlap=False: every frame is
        rendered and then gathered (one frame's latency); overlap=True: the
        gather of frame k runs under the render of frame k+1 (throughput of a stream of frames)."""
        scene_name, W, H, depth, cfg_note = WORKLOADS[workload]
        if size:
            W = H = size
        host = HostScene.named(scene_name)
        renderer = Renderer(host, device=local_rank)
        if args.tile_z:
            renderer.set_option("tile_z", args.tile_z)
        if args.block_threads:
            renderer.set_option("block_threads", args.block_threads)
        for kv in args.option:
            k, v = kv.split("=")
            renderer.set_option(k, int(v))
        direct = transport == "direct"
        shared = None
        if direct:                 # rank 0's image, mapped by every other rank (collective: raises on every rank or on none)
            shared = [SharedImage(W, H, dev)]
            shared_images.append(shared[0])
            if overlap:            # a stream of frames alternates between two images
                shared.append(SharedImage(W, H, dev))
                shared_images.append(shared[1])

        def make_pipe(bounds=None, chunks=1):
            if direct:
                return DirectStrips(shared, world, rank, dev, bounds=bounds, overlap=overlap,
                                    render_ptr=lambda address, a, b: renderer.render_device(W, H, depth, a, b, address, stream))
            pp = StripPipeline(W, H, world, rank, dev, render=None, overlap=overlap,
                               force_gather=args.force_dist, bounds=bounds, chunks=chunks, align=16)
            x0_, x1_ = pp.x0, pp.x1
            # whole strip, or (chunked) columns [a, b) of it into the matching part of the strip buffer
            pp.render = lambda buf, a=None, b=None: renderer.render_device(
                W, H, depth, x0_ if a is None else a, x1_ if b is None else b, buf.data_ptr(), stream)
            return pp

        # chunks: given, or decided with the partition below (automatic); one GPU: only if asked for
        chunks = args.chunks if (args.chunks > 0 and not overlap and not direct) else 1
        pipe = make_pipe(chunks=chunks if world == 1 else 1)     # N > 1: the measuring frames run unchunked equal strips

        def fence():
            pipe.drain()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(dev)

        # N > 1: the first warm-up steps run the equal partition and measure what a balanced one
        # needs -- every rank's kernel time and the time of the gather alone -- then the strips are
        # re-cut (tilecoderaytracer_amd.distributed.balanced_bounds) and the warm-up continues.
        partition_note = None
        warm_left = warmup
        if world > 1 and args.partition != "balanced":
            pipe = make_pipe(chunks=chunks)
        if world > 1 and args.partition == "balanced":
            pipe.step()
            fence()
            renderer.reset_timing()
            pipe.step()
            fence()
            tm = renderer.timing()
            my_kernel_ms = tm.sum_kernel_ms / max(tm.launches, 1)
            bounds = None
            agreed_chunks = None
            try:
                if direct:                   # no transfer to weigh: strips of equal measured kernel time (a peer's includes its link)
                    bounds, partition_note = balance_direct(W, my_kernel_ms, dev)
                    agreed_chunks = 1
                else:
                    bounds, partition_note, agreed_chunks = measure_and_balance(
                        pipe, W, my_kernel_ms, fence, dev, overlap=overlap,
                        chunks=0 if (args.chunks == 0 and not overlap) else chunks)
            except Exception as e:                                         # never lose the run to the tuning step
                partition_note = f"balanced partition unavailable ({e!r}); equal strips"
            # all ranks take the new strips and chunk count, or none does: `chunks` is the same number on every rank either way
            ok = torch.tensor([1 if bounds is not None else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 1:
                chunks = agreed_chunks
                pipe = make_pipe(bounds, chunks)
            else:
                if bounds is not None:
                    partition_note = "balanced partition failed on another rank; equal strips"
                pipe = make_pipe(chunks=chunks)            # what the command line asked for (or none), on every rank
            warm_left = max(warmup - 2, 1)       # at least one untimed frame on the final partition (first use of the links)
        x0, x1 = pipe.x0, pipe.x1
        # N > 1, one launch per strip: where this rank's strip starts handing out its tile rows is learned from one counting frame
        # and a few timed ones (rt_learn_tile_order; scheduling only, the pixels are the same; never loses the run)
        learned = None
        if world > 1 and chunks == 1 and not args.no_learn:
            if x1 > x0:
                try:
                    renderer.learn_tile_order(W, H, depth, x0, x1)
                    learned = True
                except Exception as e:
                    learned = f"unavailable ({e!r})"
            fence()                                   # a rank with an empty strip learns nothing but meets the others' barrier

        for _ in range(warm_left):
            pipe.step()
        fence()
        renderer.reset_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        fence()
        elapsed = time.perf_counter() - t0
        tm = renderer.timing()
        # a chunked strip is several launches per frame: the roofline pairs ONE launch's average duration with ONE
        # launch's share of the strip's pixels; the frame's kernel time is the sum over its launches
        own_kernel_ms = tm.sum_kernel_ms / max(tm.launches, 1)       # this rank's average launch
        launches_per_frame = tm.launches / max(steps, 1)
        frame_kernel_ms = tm.sum_kernel_ms / max(steps, 1)
        max_frame_kernel_ms = frame_kernel_ms
        if use_dist:
            t = torch.tensor([elapsed, frame_kernel_ms], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, max_frame_kernel_ms = float(t[0]), float(t[1])
        return dict(scene_name=scene_name, W=W, H=H, depth=depth, cfg_note=cfg_note, host=host, renderer=renderer,
                    x0=x0, x1=x1, strip=pipe.strip, pipe=pipe, transport=transport, elapsed=elapsed, kernel_ms=own_kernel_ms, max_kernel_ms=max_frame_kernel_ms,
                    launches_per_frame=launches_per_frame, frame_kernel_ms=frame_kernel_ms,
                    partition=pipe.describe(),
                    partition_note=(partition_note or "") + ("; tile rows of every strip start where rt_learn_tile_order measured best" if learned is True
                                                             else (f"; rt_learn_tile_order {learned}" if learned else "")) or None)

    def measure_transports(workload, steps, warmup, size=0):
        """(headline, other, note).  A dist run can deliver the strips to rank 0 in two ways (--transport): `steps` timed steps of
        each, the faster is the headline and the other is reported beside it.  Every rank takes the same decision: the elapsed
        times are the all-reduced maxima.  One GPU without --force-dist: one plain measurement."""
        if not use_dist:
            return measure(workload, steps, warmup, size), None, None
        wanted = ["direct"] if args.backend == "gloo" else (["rccl", "direct"] if args.transport == "auto" else [args.transport])
        done, note = [], None
        for t in wanted:
            try:
                done.append(measure(workload, steps, warmup, size, transport=t))
            except RuntimeError as e:
                # the shared image could not be created or mapped: SharedImage raises on every rank, or on none
                if t == "direct" and len(wanted) > 1 and "cannot be shared" in str(e):
                    note = f"direct transport unavailable: {e}"
                    continue
                raise
        done.sort(key=lambda r: r["elapsed"])
        first, other = done[0], (done[1] if len(done) > 1 else None)
        # The multi-GPU path's own parity property, measured in this run and outside the timed regions: the image the strips ended
        # up in on rank 0 is, bit for bit, what ONE GPU renders as one frame (dx = x / W uses the global x; the one-GPU
        # frame is what tests/test_parity_gpu.py compares with the oracle).  Correctness before speed: a faster transport whose
        # image differs does not become the headline while the other one's is right (rank 0 checks, every rank learns the decision).
        swap = 0
        if rank == 0:
            try:
                whole = torch.empty((first["W"], first["H"], 3), dtype=torch.float32, device=dev)
                first["renderer"].render_device(first["W"], first["H"], first["depth"], 0, first["W"], whole.data_ptr(), stream)
                torch.cuda.synchronize(dev)
                for r in (first, other):
                    if r is None:
                        continue
                    gathered = r["pipe"].image(r["W"])
                    if os.environ.get("TCRT_BENCH_CORRUPT") == r["transport"]:       # testing aid: spoil one pixel of that transport's image
                        gathered[0, 0, 0] += 1.0
                    differ = int((gathered[:r["W"]].view(torch.int32) != whole.view(torch.int32)).any(dim=2).sum())      # bit patterns
                    r["image_check"] = {"pixels_compared": r["W"] * r["H"], "pixels_differing": differ, "identical": differ == 0}
                del whole
            except Exception as e:
                first["image_check"] = {"error": repr(e)}
            ok_first = first.get("image_check", {}).get("identical", True)
            ok_other = other is not None and other.get("image_check", {}).get("identical", False)
            swap = 1 if (not ok_first and ok_other) else 0
        decision = torch.tensor([swap], dtype=torch.int32, device=cdev)
        dist.broadcast(decision, src=0)
        if int(decision[0]) == 1:
            first, other = other, first
            note = ((note + "; ") if note else "") + (
                f"the {other['transport']} transport was faster but its image differs from one GPU's frame: {first['transport']} is the headline")
        return first, other, note

    def transport_words(r):
        if r["transport"] == "direct":
            return (" (single frame: every frame is whole in rank 0's HBM before the next starts)")
        return ", RCCL transfers to rank 0 (single frame: every frame is delivered before the next starts)"

    def other_transport(r, steps):
        """the transport that did not become the headline, in short"""
        if r is None:
            return None
        return {"transport": r["transport"], "value": round(r["W"] * r["H"] * steps / r["elapsed"] / 1e6, 3), "unit": "Mrays/s",
                "ms_per_step": round(r["elapsed"] / steps * 1e3, 4), "slowest_rank_frame_kernel_ms": round(r["max_kernel_ms"], 4),
                "partition": r["partition"] + transport_words(r), "partition_note": r["partition_note"]}

    m, m_other, transport_note = measure_transports(args.workload, args.steps, args.warmup, args.size)

    scene_name, W, H, depth, cfg_note = m["scene_name"], m["W"], m["H"], m["depth"], m["cfg_note"]
    host, renderer, x0, x1, strip = m["host"], m["renderer"], m["x0"], m["x1"], m["strip"]
    elapsed, kernel_ms = m["elapsed"], m["kernel_ms"]

    # the north star also asks for the sphere-grid scene at every GPU count: a short second
    # measurement of configs[2] with the same partition, reported next to the headline value
    grid = None
    if args.workload == "builtin" and not args.size and not args.no_extra:
        g_steps = max(3, args.steps // 5)
        g, g_other, _ = measure_transports("grid32", g_steps, 4 if world > 1 else 2)
        grid = {
            "workload": f"{g['scene_name']} scene, {g['W']}x{g['H']}, max depth {g['depth']}",
            "baseline_config": g["cfg_note"],
            "value": round(g["W"] * g["H"] * g_steps / g["elapsed"] / 1e6, 3),
            "unit": "Mrays/s",
            "steps": g_steps,
            "ms_per_step": round(g["elapsed"] / g_steps * 1e3, 4),
            "kernel_ms": round(g["kernel_ms"], 4),
            "partition": g["partition"] + (transport_words(g) if use_dist else ""),
            "mode": ("single frame: the kernels store into rank 0's image" if g["transport"] == "direct" else "single frame: render, then gather")
                    if world > 1 else "one launch per frame",
        }
        # the north star's own wording: the sphere-grid figure "as absolute numbers and as fraction of the HBM-write roofline"
        # (12 B per pixel over the whole job's step time, against the N GPUs' aggregate HBM peak)
        grid_gbs = BYTES_PER_PIXEL * g["W"] * g["H"] * g_steps / g["elapsed"] / 1e9
        grid["hbm_write_roofline"] = {"achieved": round(grid_gbs, 3), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                      "frac": round(grid_gbs / (HBM_PEAK_GBS * world), 6)}
        if use_dist:
            grid["transport"] = g["transport"]
            grid["gathered_image_vs_one_gpu_frame"] = g.get("image_check")
            grid["other_transport"] = other_transport(g_other, g_steps)
            if g_other is not None:
                grid["other_transport"]["gathered_image_vs_one_gpu_frame"] = g_other.get("image_check")
    # N > 1: the throughput of a STREAM of frames (gather of frame k under the render of frame k+1),
    # next to the single-frame headline; a different figure, labelled as such
    pipelined, pm = None, None
    if world > 1 and not args.no_pipelined:
        try:
            pm = measure(args.workload, args.steps, max(args.warmup, 3), args.size, overlap=True, transport=m["transport"])
        except RuntimeError as e:
            if "cannot be shared" not in str(e):             # (SharedImage: raised on every rank or on none)
                raise
            pm = None
            pipelined = {"error": f"second shared image unavailable: {e}"}
    if pm is not None:
        pipelined = {
            "what": ("stream of independent frames: frame k+1 is rendered into a second shared image while the other ranks finish frame k "
                     "(two images on rank 0, a frame's all-reduce waited for two frames later)" if m["transport"] == "direct" else
                     "stream of independent frames: RCCL gather of frame k overlapped with the render of frame k+1 (two strip buffers)")
                    + "; NOT the single-frame figure `value` reports",
            "transport": m["transport"],
            "value": round(pm["W"] * pm["H"] * args.steps / pm["elapsed"] / 1e6, 3),
            "unit": "Mrays/s",
            "ms_per_step": round(pm["elapsed"] / args.steps * 1e3, 4),
            "partition": pm["partition"],
            "partition_note": pm["partition_note"],
        }

    if rank == 0:
        li = renderer.launch_info()
        launches_per_frame = m["launches_per_frame"] or 1.0
        pixels_per_launch = (x1 - x0) * H / launches_per_frame       # a strip sent in K chunks is K launches per frame
        achieved = BYTES_PER_PIXEL * pixels_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_source, compute = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and world == 1 and not args.size:
            try:
                with open(tpath) as f:
                    prof = json.load(f)
                rec = prof.get(args.workload, {})
                same = prof.get("kernel_source_sha256") == kernel_source_digest()
                traffic_source = {
                    "file": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this bench command, scripts/profile_all.sh)",
                    "profiled_kernel_source_sha256": prof.get("kernel_source_sha256"),
                    "this_build_kernel_source_sha256": kernel_source_digest(),
                    "same_kernel_source": same,
                }
                if same:
                    traffic = rec.get("hbm_bytes_per_launch")
                    if rec.get("sq_insts_valu"):
                        compute = valu_issue_roofline(rec, kernel_ms)
                else:
                    traffic_source["stale_hbm_bytes_per_launch"] = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec at 4096x4096, reflection depth 4; max per-channel delta vs CPU ref",
            "value": round(W * H * args.steps / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (hard-coded reference scene / closed-form sphere grid; no files)",
            "config": {
                "workload": f"{scene_name} scene, {W}x{H}, max depth {depth}",
                "baseline_config": cfg_note,
                "objects": host.object_count,
                "partition": m["partition"] + (transport_words(m) if use_dist else ""),
                "partition_note": m["partition_note"],
                "block_threads": li.block_threads,
                "lds_bytes_per_block": li.lds_bytes,
                "wave_tile": f"{li.tile_x}x{li.tile_z}",
                "max_delta_vs_cpu_ref": None,
                "parity": "not measured in this run (every pixel of this frame is compared with the oracle in tests/test_parity_gpu.py)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": li.kernel.decode() if isinstance(li.kernel, bytes) else str(li.kernel),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "compute": compute,
                "kernel_ms": round(kernel_ms, 4),
                "launches_per_frame": round(launches_per_frame, 3),
                "frame_kernel_ms": round(m["frame_kernel_ms"], 4),
                "algorithmic_bytes_per_launch": round(BYTES_PER_PIXEL * pixels_per_launch, 1),
                "note": "12 B/pixel framebuffer store is the only HBM traffic that scales with the image; "
                        "the kernel is bound by fp32 (non-FMA) instruction issue, see DESIGN.md",
            },
        }
        if use_dist:
            out["config"]["transport"] = m["transport"]
            out["config"]["other_transport"] = other_transport(m_other, args.steps)
            if transport_note:
                out["config"]["transport_note"] = transport_note
            out["config"]["gathered_image_vs_one_gpu_frame"] = m.get("image_check")       # (measured before the choice, measure_transports())
            if m_other is not None:
                out["config"]["other_transport"]["gathered_image_vs_one_gpu_frame"] = m_other.get("image_check")
        if world > 1:
            # what bounds a single frame on N GPUs, in the line itself: every peer's columns cross ONE xGMI link into rank 0
            peer_mb = BYTES_PER_PIXEL * W * H * (world - 1) / world / 1e6
            out["config"]["scaling_note"] = (
                f"single-frame delivery moves {peer_mb:.0f} MB ({world - 1} strip(s) of 12 B/pixel) into rank 0 per frame, each peer over its own "
                f"point-to-point xGMI link; frame kernel time of the slowest rank {m['max_kernel_ms']:.3f} ms of {elapsed / args.steps * 1e3:.3f} ms per step: "
                "where the strips render faster than their columns travel (the built-in scene: 0.6 ms of rendering in all) the figure is "
                "transfer-bound whatever the partition -- `sphere_grid` is the north star's scaling scene, `pipelined` the throughput "
                "with the transfers hidden")
        if grid is not None:
            out["sphere_grid"] = grid
        if pipelined is not None:
            out["pipelined"] = pipelined
        if world == 1 and not args.no_extra:
            # SURVEY.md 8(d) secondary figures, from the counting build at 1024 x 1024 (per-pixel counts
            # barely depend on the resolution): all rays traced, and intersection tests against the
            # non-packed, non-FMA fp32 VALU peak (CUs x 64 lanes x clock; ~15 flop per test)
            try:
                _, st = renderer.render_stats(1024, 1024, depth)
                px = 1024.0 * 1024.0
                rays_pp = (st["nearest_rays"] + st["shadow_rays"]) / px
                tests_pp = (st["wave_sphere_tests"] + st["wave_plane_tests"] + st["wave_box_tests"]) * 64.0 / px
                pixels_per_s = out["value"] * 1e6
                out["secondary"] = {
                    "rays_per_pixel": round(rays_pp, 3),
                    "total_rays_per_s": round(pixels_per_s * rays_pp, 0),
                    "tests_issued_per_pixel": round(tests_pp, 2),
                    "tests_per_s": round(pixels_per_s * tests_pp, 0),
                    "test_flop_frac_of_valu_peak": round(pixels_per_s * tests_pp * 15.0 / VALU_PEAK_LANE_OPS, 4),
                    "note": "counting build, 1024x1024, ~15 flop per test; peak = 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz = 78.6 T "
                            "non-fused fp32 lane-ops/s (MI355X_MICROARCH.md: 157.3 TF fp32 counts an FMA as two)",
                }
            except Exception as e:
                out["secondary_error"] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            try:
                cols = args.cpu_sample_columns or {"builtin": 4096, "builtin8k": 2048}.get(args.workload, 64)
                gpu_image = renderer.render(W, H, depth)          # host copy of the frame, outside the timed region
                out["cpu_baseline"] = cpu_baseline(scene_name, W, H, depth, cols, gpu_image)
                par = out["cpu_baseline"].get("parity")
                if par:
                    out["config"]["max_delta_vs_cpu_ref"] = par["max_abs_delta"]
                    out["config"]["parity"] = (f"measured in this run: {par['pixels_compared']} pixels against the CPU oracle, "
                                               f"{par['pixels_over_1e-6']} differ by more than 1e-6")
            except Exception as e:  # the baseline is a report, never the product
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = repr(e)
            if grid is not None:
                # "... next to the reference CPU path timed on the node's own host cores (core count stated) in the same run":
                # the oracle on the sphere-grid frame too, a bounded sample (one run of ~1 s on all cores), every sampled
                # column compared with the GPU's frame
                try:
                    grid_image = g["renderer"].render(g["W"], g["H"], g["depth"])
                    grid["cpu_baseline"] = cpu_baseline(g["scene_name"], g["W"], g["H"], g["depth"], 64, grid_image, repeats=1)
                    del grid_image
                except Exception as e:
                    grid["cpu_baseline_error"] = repr(e)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
            saved_stdout = None
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        for image in shared_images:              # the mappings first, then rank 0's allocations
            if rank != 0:
                image.close()
        dist.barrier()
        for image in shared_images:
            image.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
