import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces are built in-tree (git-ignored): build them when a checkout arrives without them
    import subprocess
    need = [os.path.join(ROOT, "phagefilter_amd", "libpfq.so"), os.path.join(ROOT, "phagefilter_amd", "phage_filter"),
            os.path.join(ROOT, "oracle", "libpfq_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "phagefilter_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


def _has_gpu() -> bool:
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.skip("no HIP device")
    return 0
