import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# device memory shared between processes (rt_shared_image_*, RCCL) goes through dmabuf on this pool's driver; the variable is
# read when a HIP runtime starts -- here, before torch brings one in, and inherited by every process the tests start
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# torch first: it carries its own copy of the HIP runtime, and a process that has already
# initialised the system one through libtcrt.so (DT_NEEDED /opt/rocm/lib/libamdhip64.so) finds
# "No HIP GPUs" when torch initialises afterwards; the other order works (seen on the GPU box
# when test_parity_gpu.py is collected alone).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is only plumbing for a few tests
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build product + oracle if the shared objects are missing (cheap no-op otherwise)."""
    need = [
        os.path.join(ROOT, "tilecoderaytracer_amd", "lib", "libtcrt.so"),
        os.path.join(ROOT, "tilecoderaytracer_amd", "lib", "libtcrt_host.so"),
        os.path.join(ROOT, "oracle", "liboracle.so"),
    ]
    if all(os.path.exists(p) for p in need):
        return
    from tilecoderaytracer_amd import build
    build.build_all()


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def have_gpu():
    import ctypes
    from tilecoderaytracer_amd import capi
    n = ctypes.c_int(0)
    rc = capi.load_library().rt_device_count(ctypes.byref(n))
    return rc == 0 and n.value > 0
