import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference decoder compiled into oracle/_ref (skip when absent)."""
    from oracle import oracle as orc
    if os.path.exists("/root/reference/pdmp3.c"):
        orc.build()
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return orc.Reference()


@pytest.fixture(scope="session")
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pdmp3_amd
    return pdmp3_amd.Engine(0)


@pytest.fixture(scope="session")
def emul():
    """Host build of the device pipeline (tests/host_emul): the kernel's own source, one wave = 64 fibers."""
    import ctypes as C
    import subprocess
    d = os.path.join(ROOT, "tests", "host_emul")
    so = os.path.join(d, "libemul.so")
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []      # (else libm's fmaf: same results, slower)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + fma +
                          ["-o", so, os.path.join(d, "emul.cpp")])
    lib = C.CDLL(so)
    lib.emul_state_floats.restype = C.c_size_t
    return lib


C2_SEED = 0x5EED0000C2
