"""The N > 1 path on CPU: world_size-2 `gloo` processes, each decoding its
frame-range shard (with the 2-frame halo) through the host build of the device
pipeline, PCM gathered to rank 0 exactly as bench.py does over RCCL, compared
with the unsharded decode and the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 46
SEED = 0x5EED0000C5


def _worker(rank, world, port, emul_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from pdmp3_amd.sharding import shard_with_halo
    first, count, halo = shard_with_halo(N_FRAMES, rank, world)
    o = Oracle()
    sp, sd = o.generate(SEED, first, count)          # counter-based: each rank makes its own shard
    em = C.CDLL(emul_path)
    pcm = np.zeros((count, 2304), np.int16)
    em.emul_decode_frames(sp.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p), count, None,
                          pcm.ctypes.data_as(C.c_void_p), None, 4)
    mine = torch.from_numpy(pcm[halo:].copy()).view(torch.uint8)     # collectives carry bytes (no int16 in gloo/RCCL)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[0]]))
    pad = int(max(int(s) for s in sizes))
    buf = torch.zeros((pad, 4608), dtype=torch.uint8)
    buf[:mine.shape[0]] = mine
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank == 0:
        whole = torch.cat([g[:int(s)] for g, s in zip(gathered, sizes)]).view(torch.int16).numpy()
        np.save(out_path, whole)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(emul, oracle, tmp_path):
    out = str(tmp_path / "gathered.npy")
    emul_path = os.path.join(ROOT, "tests", "host_emul", "libemul.so")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, emul_path, out), nprocs=2, join=True)
    got = np.load(out)
    sp, sd = oracle.generate(SEED, 0, N_FRAMES)
    whole = np.zeros((N_FRAMES, 2304), np.int16)
    emul.emul_decode_frames(sp.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p), N_FRAMES, None,
                            whole.ctypes.data_as(C.c_void_p), None, 0)
    assert got.shape == whole.shape
    assert np.array_equal(got, whole), "sharded + gathered PCM differs from the unsharded decode"
    want = oracle.decode(sp, sd)
    assert np.abs(got.astype(int) - want).max() <= 1


def test_sharding_arithmetic():
    from pdmp3_amd.sharding import frame_range, shard_with_halo, assign_files
    for n in (1, 7, 2048, 1000000):
        for w in (1, 2, 3, 8):
            spans = [frame_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            for r in range(w):
                first, cnt, halo = shard_with_halo(n, r, w)
                assert first == spans[r][0] - halo and cnt == spans[r][1] - spans[r][0] + halo
                assert halo == min(2, spans[r][0])
    a = assign_files([5, 9, 1, 7, 3, 3], 3)
    assert sorted(sum(a, [])) == list(range(6))
    loads = [sum([5, 9, 1, 7, 3, 3][i] for i in g) for g in a]
    assert max(loads) - min(loads) <= 3
