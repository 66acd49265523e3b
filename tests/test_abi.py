"""The C-ABI library loads and exports every symbol include/pdmp3_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pdmp3_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports():
    import pdmp3_amd
    pdmp3_amd.build_library()
    lib = pdmp3_amd.load_library()
    names = _declared("pdmp3_hip.h")
    assert "pdmp3_hip_decode_frames" in names and "pdmp3_hip_create" in names
    for n in names:
        assert hasattr(lib, n), "include/pdmp3_hip.h declares %s but the library lacks it" % n
    assert lib.pdmp3_hip_state_bytes() > 0


def test_node_library_exports_and_shard_arithmetic():
    """include/pdmp3_node.h: every declared symbol is in libpdmp3_hip.so (RCCL itself is looked up at run time: the
    library must load without it), and pdmp3_node_shard -- a pure function -- is pdmp3_amd.sharding.shard_with_halo:
    even and uneven splits, and cuts after runs of mono frames, RESET frames included"""
    import pdmp3_amd
    from pdmp3_amd import hip, sharding
    lib = pdmp3_amd.load_library()
    names = _declared("pdmp3_node.h")
    assert set(hip.NODE_EXPORTS) == set(n for n in names if n.startswith("pdmp3_node_"))
    for n in names:
        assert hasattr(lib, n), "include/pdmp3_node.h declares %s but the library lacks it" % n
    deps = os.popen("ldd %s" % pdmp3_amd.library_path()).read()
    assert "rccl" not in deps and "nccl" not in deps, "the engine library must not link against RCCL"
    rng = np.random.default_rng(5)
    for n, world in ((1000000, 8), (10, 3), (7, 8), (1, 1), (125000, 2), (4097, 5)):
        for rank in range(world):
            assert hip.node_shard(n, rank, world) == sharding.shard_with_halo(n, rank, world), (n, rank, world)
    for trial in range(300):
        n = int(rng.integers(1, 120))
        world = int(rng.integers(1, 9))
        # runs of stereo (mode 1) and mono (mode 3) frames, RESET flags sprinkled in
        flags = np.zeros(n, np.uint8)
        f = 0
        while f < n:
            run = int(rng.integers(1, 12))
            flags[f:f + run] = (3 if rng.integers(0, 2) else 1) << 2
            f += run
        flags[rng.random(n) < 0.06] |= 0x40
        for rank in range(world):
            assert hip.node_shard(n, rank, world, flags) == sharding.shard_with_halo(n, rank, world, frame_flags=flags), (trial, rank, world)


def test_api_library_exports():
    """libpdmp3.so: every function include/pdmp3.h (the reference's API) and include/pdmp3_bulk.h declare"""
    from pdmp3_amd import api
    lib = api.load_library()
    names = [n for h in ("pdmp3.h", "pdmp3_bulk.h") for n in _declared(h) if not n.startswith("pdmp3_hip_")]
    names.append("pdmp3")                          # the CLI driver itself (pdmp3.c:2540)
    assert set(api.API_EXPORTS) <= set(names) and set(api.BULK_EXPORTS) <= set(names)
    for n in names:
        assert hasattr(lib, n), "include/ declares %s but libpdmp3.so lacks it" % n


def test_side_record_layout():
    from pdmp3_amd.hip import SIDE_DTYPE
    assert SIDE_DTYPE.itemsize == 128
    assert SIDE_DTYPE.fields["scalefac_l"][1] == 8 and SIDE_DTYPE.fields["scalefac_s"][1] == 30


def test_host_generator_matches_oracle(oracle):
    from pdmp3_amd import hip
    sp, sd = hip.host_generate(0x5EED0000C2, 5, 24)
    sp2, sd2 = oracle.generate(0x5EED0000C2, 5, 24)
    assert np.array_equal(sp, sp2)
    assert np.array_equal(sd.view(np.uint8), sd2.view(np.uint8))


def test_no_gpu_means_loud_failure():
    import torch
    import pytest
    import pdmp3_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pdmp3_amd.Engine(0)
    lib = pdmp3_amd.load_library()
    h = C.c_void_p()
    assert lib.pdmp3_hip_create(0, C.byref(h)) != 0
    assert b"hipSetDevice" in lib.pdmp3_hip_last_error()


def test_corpus_assign_in_c_equals_sharding_assign_files():
    """pdmp3_amd_corpus_assign (include/pdmp3_bulk.h, the dealer of pdmp3_amd_corpus_decode) == sharding.assign_files: largest
    first, ties to the lower slot -- 300 random corpora; no GPU"""
    import random
    from pdmp3_amd import api
    from pdmp3_amd.sharding import assign_files
    rnd = random.Random(7)
    for _ in range(300):
        n, w = rnd.randint(1, 40), rnd.randint(1, 9)
        sizes = [rnd.choice([100, 100, 5000, rnd.randint(1, 10 ** 7)]) for _ in range(n)]
        want = [None] * n
        for k, g in enumerate(assign_files(sizes, w)):
            for i in g:
                want[i] = k
        assert api.corpus_assign(sizes, w) == want
