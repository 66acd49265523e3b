"""The persistent granule kernel (k_decode_p) on the GPU: tests/ring_variant_checks.py in a child process that loads the
variant library built with -DPDMP3_WITH_RING_KERNEL (test it or delete it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_ring_variant_library():
    lib = os.path.join(ROOT, "pdmp3_amd", "libpdmp3_hip_ring.so")
    assert os.path.exists(lib), "build it: make -C pdmp3_amd/csrc (or __graft_entry__.build())"
    env = dict(os.environ, PDMP3_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "ring_variant_checks.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout and "failed" not in r.stdout, tail
