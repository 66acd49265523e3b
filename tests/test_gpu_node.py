"""include/pdmp3_node.h on the ONE GPU of the test box: the C-ABI form of the sharded decode.

What one GPU can verify: RCCL comes up from inside the library (ncclCommInitAll over one device, looked up with dlopen)
and a one-rank node decodes like the engine; the shard arithmetic, the halos (fixed, and moved back past mono runs), the
per-rank host threads and the gather's bookkeeping run with 2, 3 and 5 ranks that all sit on device 0 and exchange by
device copies (PDMP3_NODE_COPY: RCCL refuses a device that is listed twice -- which is also tested).  The exchange over
xGMI between different GPUs has never run anywhere this repository was developed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED_C5 = 0x5EED0000C5


def _unsharded(engine, n):
    import torch
    sp, sd, pcm = engine.alloc_frames(n)
    engine.generate(SEED_C5, 0, n, sp, sd)
    engine.decode(sp, sd, pcm)
    torch.cuda.synchronize()
    return pcm


def test_one_rank_over_rccl_equals_the_engine(engine):
    import torch
    from pdmp3_amd.hip import NodeDecoder, NODE_RCCL
    n = 20000
    node = NodeDecoder([0], NODE_RCCL)
    try:
        pcm, tm = node.decode_generated(SEED_C5, n)
        assert tm.rccl_ranks == 1 and tm.gather_bytes == 0 and tm.decode_ms > 0
        assert torch.equal(pcm, _unsharded(engine, n))
    finally:
        node.close()


def test_rccl_refuses_a_device_twice_and_says_so():
    from pdmp3_amd.hip import NodeDecoder, NODE_RCCL
    with pytest.raises(RuntimeError, match="PDMP3_NODE_COPY"):
        NodeDecoder([0, 0], NODE_RCCL)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_generated_stream_in_shards_on_one_gpu(engine, world):
    """C5's shape at a size the test box decodes in milliseconds: `world` ranks, all on device 0, each generating and
    decoding its shard from the 2-frame halo on a host thread of its own; gathered PCM == the unsharded decode, bit for bit"""
    import torch
    from pdmp3_amd.hip import NodeDecoder, NODE_COPY
    n = 30000 + world                                        # (an uneven split)
    node = NodeDecoder([0] * world, NODE_COPY)
    try:
        pcm, tm = node.decode_generated(SEED_C5, n)
        lo1 = n // world + (1 if n % world else 0)
        assert tm.rccl_ranks == 0 and tm.gather_bytes == (n - lo1) * 4608
        assert torch.equal(pcm, _unsharded(engine, n))
        # round 6: the shard goes in slices (state carried from slice to slice), a slice's PCM leaves while the next decodes
        assert tm.slices == max(s for s in range(1, 9) if s == 1 or (lo1 + 2) // s >= 4096), tm.slices
        assert tm.total_ms > 0 and tm.decode_ms > 0 and tm.gather_ms > 0
        pcm2, _ = node.decode_generated(SEED_C5, 777)        # the node's buffers are kept and reused: a smaller stream after a larger one
        assert torch.equal(pcm2, _unsharded(engine, 777))
    finally:
        node.close()


@pytest.mark.parametrize("slices", [1, 2, 5, 8])
def test_sliced_exchange_is_the_unsliced_one(engine, slices, monkeypatch):
    """$PDMP3_NODE_SLICES: any number of pieces per shard, same PCM (3 ranks on one GPU, 50 001 frames: pieces of >= 4096
    frames, the last one shorter, the first one of ranks 1 and 2 starting behind the halo)"""
    import torch
    from pdmp3_amd.hip import NodeDecoder, NODE_COPY
    monkeypatch.setenv("PDMP3_NODE_SLICES", str(slices))
    n = 50001
    node = NodeDecoder([0] * 3, NODE_COPY)
    try:
        pcm, tm = node.decode_generated(SEED_C5, n)
        assert tm.slices == min(slices, 4)                   # 16 669 frames per shard: at most 4 pieces of >= 4096
        assert torch.equal(pcm, _unsharded(engine, n))
    finally:
        node.close()


def test_two_gpus_over_rccl_equals_the_engine(engine):
    """the documented product path, as soon as a box has two GPUs: NodeDecoder([0, 1], NODE_RCCL) against the
    unsharded decode -- and the caller's current device is still device 0 afterwards.  Skipped on the one-GPU test box:
    until this has passed on hardware the RCCL transport with more than one rank is EXPERIMENTAL (include/pdmp3_node.h)."""
    import torch
    from pdmp3_amd.hip import NodeDecoder, NODE_RCCL
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: the exchange between different devices cannot run")
    n = 250000
    torch.cuda.set_device(0)
    node = NodeDecoder([0, 1], NODE_RCCL)
    try:
        pcm, tm = node.decode_generated(SEED_C5, n)
        assert torch.cuda.current_device() == 0
        assert tm.rccl_ranks == 2 and tm.gather_bytes == (n // 2) * 4608 and tm.slices == 8
        assert torch.equal(pcm, _unsharded(engine, n))
    finally:
        node.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_records_of_a_mode_switching_stream_in_shards(engine, world):
    """records in host memory, stereo / mono / stereo runs: the cuts that follow mono frames start in front of the last
    stereo frame (pdmp3_node_shard reads the records' own flag bytes); == the whole decode"""
    import torch
    from pdmp3_amd import api
    from pdmp3_amd.hip import NodeDecoder, NODE_COPY
    from pdmp3_amd.packer import packer
    parts = [dict(n_frames=19, seed=41, bitrate_index=9), dict(n_frames=23, seed=42, mode=3, bitrate_index=7),
             dict(n_frames=3, seed=43, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10)),
             dict(n_frames=17, seed=44, mode=3, bitrate_index=7),
             dict(n_frames=21, seed=45, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10))]
    mp3 = b"".join(packer.generate(**p) for p in parts)
    sp, sd = api.parse_like_cli(mp3, 64)
    n = sp.shape[0]
    dsp, dsd = engine.upload(sp, sd)
    whole = torch.zeros((n, 2304), dtype=torch.int16, device=engine.tdev)
    engine.decode(dsp, dsd, whole)
    torch.cuda.synchronize()
    node = NodeDecoder([0] * world, NODE_COPY)
    try:
        pcm, tm = node.decode_records(sp, sd)
        assert torch.equal(pcm, whole)
    finally:
        node.close()
