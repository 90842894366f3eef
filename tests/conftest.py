import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "cuda-pathtracer_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the checker (oracle/) and the product library once per session if missing."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "libptmi_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "cuda-pathtracer_amd", "libptmi.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuda-pathtracer_amd"), "-j4"])   # hipcc cross-compiles gfx950 anywhere
    yield
