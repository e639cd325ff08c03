import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gs-sr_amd"), os.path.dirname(os.path.abspath(__file__)), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built libraries are git-ignored: build them when a fresh checkout runs the suite (hipcc cross-compiles on CPU)
    import shutil
    import subprocess
    hip_so = os.path.join(ROOT, "gs-sr_amd", "libgsrast_hip.so")
    if not os.path.exists(hip_so) and shutil.which("hipcc") or (not os.path.exists(hip_so) and os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "gs-sr_amd", "csrc"), "-j8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libgsr_oracle.so")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
