import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _ensure_built():
    """CPU-side libraries are cheap to (re)build; the HIP library is built by __graft_entry__.build()."""
    need = [os.path.join(ROOT, "pathtracer_amd", "lib", "libpthost.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "pathtracer_amd", "csrc"), "host"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


_ensure_built()


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("scenes"))


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "assets"))
