"""`python bench.py --gpus N` with no rank variables in the environment starts its own N ranks
(bench.self_launch): as a child process, before the parent has imported torch or anything that
touches HIP, relaying rank 0's single JSON line and the child's exit code.  The reference's main()
spawns its ranks itself too (src/RayTracer.cpp:1536-1566).  No GPU is needed: a stub stands in for
`python -m torch.distributed.run` (TCRT_BENCH_LAUNCHER) and records what it was started with."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

STUB = r'''
import json, os, sys
rec = {"argv": sys.argv[1:], "env": {k: os.environ.get(k) for k in
       ("TCRT_BENCH_CHILD", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}}
json.dump(rec, open(os.environ["TCRT_STUB_OUT"], "w"))
mode = os.environ.get("TCRT_STUB_MODE", "ok")
print("RCCL version banner (noise on stdout)")
if mode == "fail":
    print("rank 1 died", file=sys.stderr)
    sys.exit(3)
print(json.dumps({"metric": "Mrays/sec", "value": 1.5, "n_gpus": 2}))
if mode == "two":
    print(json.dumps({"metric": "Mrays/sec", "value": 2.5, "n_gpus": 2}))
print("{not json")
print(json.dumps({"no_metric_here": 1}))
'''


def run_bench(tmp_path, args, mode="ok", extra_env=None, python_flags=()):
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(STUB)
    out = tmp_path / "stub_out.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TCRT_BENCH_CHILD")}
    env.update({"TCRT_BENCH_LAUNCHER": f"{sys.executable} {stub}", "TCRT_STUB_OUT": str(out), "TCRT_STUB_MODE": mode})
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, *python_flags, BENCH, *args], env=env, capture_output=True, text=True, timeout=120)
    rec = json.load(open(out)) if out.exists() else None
    return r, rec


def test_launch_command_and_environment():
    sys.path.insert(0, ROOT)
    import bench
    os.environ.pop("TCRT_BENCH_LAUNCHER", None)
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20"], 29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    k = cmd.index(os.path.abspath(BENCH))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20"]            # the caller's own arguments, unchanged
    env = bench.launch_environment({"RANK": "3", "WORLD_SIZE": "4", "LOCAL_RANK": "3", "MASTER_PORT": "1", "PATH": "/bin"})
    assert env["TCRT_BENCH_CHILD"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert not any(k in env for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"))
    assert bench.launch_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # the caller's choice stands
    p1, p2 = bench.free_port(), bench.free_port()
    assert 1024 < p1 < 65536 and 1024 < p2 < 65536


def test_result_lines_keep_bench_lines_only():
    sys.path.insert(0, ROOT)
    import bench
    found, noise = bench.result_lines('banner\n{"metric": "m", "value": 1}\n{"x": 1}\n{broken\n')
    assert found == ['{"metric": "m", "value": 1}'] and noise == ["banner", '{"x": 1}', "{broken"]


def test_parent_starts_the_ranks_and_relays_one_json_line(tmp_path):
    r, rec = run_bench(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "grid32"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "Mrays/sec", "value": 1.5, "n_gpus": 2}
    assert "RCCL version banner" in r.stderr and "no_metric_here" in r.stderr      # the child's other output is not lost
    # the child: the launcher's arguments, then this file with the caller's own arguments
    a = rec["argv"]
    assert a[:2] == ["--nnodes=1", "--nproc-per-node=2"] and a[2:4] == ["--master-addr", "127.0.0.1"] and a[4] == "--master-port"
    assert int(a[5]) > 1024 and a[6] == os.path.abspath(BENCH)
    assert a[7:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "grid32"]
    assert rec["env"]["TCRT_BENCH_CHILD"] == "1" and rec["env"]["RANK"] is None and rec["env"]["WORLD_SIZE"] is None
    assert rec["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_parent_touches_no_gpu_module_before_the_spawn(tmp_path):
    """-X importtime lists every module the PARENT imports (the flag is not inherited by the child): neither torch nor
    the package that loads libtcrt.so / libamdhip64.so may be among them -- a process that has initialised the GPU must
    not start the ranks (and must never exec)."""
    r, rec = run_bench(tmp_path, ["--gpus", "2"], python_flags=("-X", "importtime"))
    assert r.returncode == 0 and rec is not None
    imported = [line.rsplit("|", 1)[-1].strip() for line in r.stderr.splitlines() if line.startswith("import time:")]
    assert "json" in imported and "subprocess" in imported                  # the listing works
    bad = [m for m in imported if m.split(".")[0] in ("torch", "tilecoderaytracer_amd")]
    assert not bad, bad
    text = open(BENCH).read()
    assert "os.exec" not in text and "execv" not in text


def test_child_failure_is_the_parents_exit_code(tmp_path):
    r, rec = run_bench(tmp_path, ["--gpus", "2"], mode="fail")
    assert r.returncode == 3 and r.stdout == "" and "rank 1 died" in r.stderr


def test_more_than_one_result_line_is_an_error(tmp_path):
    r, rec = run_bench(tmp_path, ["--gpus", "2"], mode="two")
    assert r.returncode == 1 and r.stdout == ""


@pytest.mark.parametrize("env", [{"RANK": "0", "WORLD_SIZE": "1"}, {"TCRT_BENCH_CHILD": "1"}])
def test_a_rank_never_starts_ranks(tmp_path, env):
    """Inside a launcher's rank (or a child of this file) a WORLD_SIZE that does not match --gpus is an error, not a second generation."""
    r, rec = run_bench(tmp_path, ["--gpus", "2"], extra_env=env)
    assert r.returncode != 0 and rec is None and "does not match" in r.stderr
