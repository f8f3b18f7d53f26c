import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE = "/root/reference"
REF_HUMANOID_XML = os.path.join(REFERENCE, "simulation/mujoco/model/humanoid/humanoid.xml")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import subprocess
    lib = os.path.join(ROOT, "humanoid_mujoco_amd", "libhb.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", ROOT, "-s", "humanoid_mujoco_amd/libhb.so"])
    from oracle_lib import build_oracle
    build_oracle()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


@pytest.fixture(scope="session")
def hbmod():
    import humanoid_mujoco_amd as hb
    return hb


@pytest.fixture()
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def humanoid_model(hbmod):
    from oracle_lib import HUMANOID_HBM
    return hbmod.Model.load(HUMANOID_HBM)


def gpu_count():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) != 0:
            return 0
        return n.value
    except OSError:
        return 0


@pytest.fixture(scope="session")
def gpu(hbmod):
    if gpu_count() < 1:
        pytest.skip("no GPU visible")
    return 0
