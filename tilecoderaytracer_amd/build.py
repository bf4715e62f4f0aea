"""Build every native piece: the HIP library, the host library/executable and
(test infrastructure) the oracle.  Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def _make(directory, *targets):
    cmd = ["make", "-C", directory, *targets]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stdout[-4000:]}")
    return r.stdout


def build_product():
    return _make(os.path.join(_HERE, "csrc"))


def build_oracle():
    return _make(os.path.join(_ROOT, "oracle"))


def build_all():
    build_product()
    build_oracle()
