"""The multi-GPU paths exercised on the ONE GPU of the test box: bench.py's N = 2 code path (two ranks sharing GPU 0,
gloo instead of RCCL -- the launch line, sharding, halo and gather are the ones the 8-GPU run uses), and a C4-style
corpus dealt over per-device decoders (pdmp3_amd_bulk_new_on), which degrade to device 0 twice here."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pdmp3_amd.packer import packer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gathered_pcm_equals_unsharded(engine, tmp_path):
    import torch
    n = 9000                                                 # (two slices of >= 4096 frames per shard: the pipelined exchange's piece arithmetic)
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, PDMP3_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--frames", str(n), "--dump-gathered", out, "--strong-frames", "24000"]
    # a fresh child process (never an exec from this process, which has initialised the GPU)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_gpu"] == n
    assert d["gather_bytes"] == n * 4608 and d["gather_ms"] > 0 and d["value"] > 0
    # round 6: the measured step is decode + exchange (sliced, overlapped on real hardware; blocking through host memory with
    # gloo here); the decode-only figure of rounds 1-5 beside it, and the fixed-stream (strong-scaling) leg
    assert d["slices"] == 2 and d["value_decode_only"] >= d["value"] and 0.0 <= d["overlap_frac"] <= 1.0
    assert d["parity"]["pipelined_gather_equals_one_piece_gather"] is True
    assert d["strong_scaling"]["frames_per_gpu"] == 12000 and d["strong_scaling"]["frames_per_s"] > 0, d["strong_scaling"]
    # the line explains itself: what each rank ran, how many ranks the collective library saw (0 here:
    # gloo stands in for RCCL on the one GPU), and the parity of the gathered PCM around the shard boundary
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all("k_decode" in r["kernel"] for r in d["ranks"])
    assert d["ranks"][1]["halo_frames"] == 2 and d["ranks"][1]["first_frame"] == n - 2
    assert d["rccl_ranks"] == 0 and d["collective_backend"] == "gloo"
    assert d["parity"]["ok"] and d["parity"]["max_abs_diff_lsb"] <= 1 and d["parity"]["frames"] == 4 + 4 + 2, d["parity"]
    got = np.load(out)
    assert got.shape == (2 * n, 2304)
    spectra, side, pcm = engine.alloc_frames(2 * n)
    engine.generate(0x5EED0000C5, 0, 2 * n, spectra, side)
    engine.decode(spectra, side, pcm)
    torch.cuda.synchronize()
    assert np.array_equal(got, pcm.cpu().numpy()), "sharded + gathered PCM differs from the unsharded decode"


def test_bench_exchange_that_hangs_still_ends_with_a_line():
    """the pipelined exchange is point-to-point traffic that only the driver's multi-GPU run ever issues over RCCL: a peer that
    never sends (PDMP3_BENCH_TEST_HANG) must not cost the run its line -- the watchdog prints the decode-only measurement,
    labelled, and every rank exits"""
    env = dict(os.environ, PDMP3_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", PDMP3_BENCH_TEST_HANG="1", PDMP3_BENCH_WATCHDOG_S="4")
    port = 29300 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "9000"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["value"] == d["value_decode_only"]
    assert "watchdog" in d["pipelined_exchange_failed"] and d["roofline"]["frac"] > 0


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it (the driver's N = 1 command with another N): bench.py
    starts the ranks itself as a fresh child torch.distributed.run and relays the one JSON line and the exit code"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PDMP3_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "3000"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["frames_per_gpu"] == 3000
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and d["parity"]["ok"] and d["value"] > 0


def test_corpus_dealt_over_devices(oracle):
    """C4's partitioning (whole files per GPU, largest first) through per-device decoders"""
    import torch
    from pdmp3_amd import api
    from pdmp3_amd.sharding import assign_files
    kinds = [dict(sfreq=0, mode=1, mode_ext=2, bitrate_index=14), dict(sfreq=1, mode=3, bitrate_index=7),
             dict(sfreq=2, mode=0, mode_ext=0, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=50),
             dict(sfreq=0, mode=2, bitrate_index=12, block_pct=(10, 10, 70, 10))]
    files = [packer.generate(n_frames=40 + 13 * k, seed=900 + k, **kinds[k % 4]) for k in range(10)]
    ndev = max(1, torch.cuda.device_count())
    devices = [0, 1 % ndev]                            # two "GPUs": the second one is device 0 again on a 1-GPU box
    groups = assign_files([len(f) for f in files], len(devices))
    decs = [api.BulkDecoder(threads=2, window_frames=64, device=d) for d in devices]
    try:
        outs = {}
        for dec, grp in zip(decs, groups):
            for i, pcm in zip(grp, dec.decode_many([files[i] for i in grp])):
                outs[i] = pcm
    finally:
        for dec in decs:
            dec.close()
    assert sorted(outs) == list(range(len(files)))
    for i, f in enumerate(files):
        want = np.frombuffer(oracle.decode_buffer_like_cli(f), dtype=np.int16)
        assert outs[i].shape == want.shape and np.abs(outs[i].astype(np.int32) - want).max() <= 1, i


def test_corpus_dealt_over_devices_from_c(oracle):
    """the same from C: pdmp3_amd_corpus_decode (include/pdmp3_bulk.h) deals the files largest first over its devices -- two
    entries, both device 0 on the test box -- with a decoder and a host thread each; an LSF file among them (PDMP3_ISO_LSF)"""
    from pdmp3_amd import api
    kinds = [dict(sfreq=0, mode=1, mode_ext=2, bitrate_index=14), dict(sfreq=1, mode=3, bitrate_index=7),
             dict(sfreq=2, mode=0, mode_ext=0, vbr=True, vbr_hi=12, block_pct=(40, 10, 40, 10), mixed_pct=50),   # (32 kHz <= 224 kbps: SURVEY H10)
             dict(sfreq=0, mode=2, bitrate_index=12, block_pct=(10, 10, 70, 10))]
    files = [packer.generate(n_frames=300 + 131 * k, seed=700 + k, **kinds[k % 4]) for k in range(9)]
    files.append(packer.generate(n_frames=200, seed=77, sfreq=2, mode=1, mode_ext=2, bitrate_index=8, version=1, iso_strict=True))
    for host_huffman in (False, True):
        outs = api.corpus_decode([0, 0], files, iso=api.ISO_LSF, threads=2, window_frames=256, host_huffman=host_huffman)
        for i, f in enumerate(files):
            want = np.frombuffer(oracle.decode_buffer_like_cli_iso(f, api.ISO_LSF), dtype=np.int16)
            assert outs[i].shape == want.shape and want.size > 0 and np.abs(outs[i].astype(np.int32) - want).max() <= 1, (i, host_huffman)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_record_shards_of_a_mode_switching_stream(engine, world):
    """frame-range shards of a stream with stereo / mono / stereo runs: each rank's first frame comes from the records'
    own flag bytes (sharding.shard_records -> halo_start: a cut after mono frames starts in front of the last stereo
    frame, where channel 1's state is from); shards decoded one by one on the GPU, halo discarded, == the whole decode"""
    import torch
    from pdmp3_amd import api
    from pdmp3_amd.sharding import shard_records
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
    out = []
    for rank in range(world):
        first, count, discard = shard_records(sd, rank, world)
        pcm = torch.zeros((count, 2304), dtype=torch.int16, device=engine.tdev)
        engine.decode(dsp[first:first + count].contiguous(), dsd[first:first + count].contiguous(), pcm)
        out.append(pcm[discard:])
    torch.cuda.synchronize()
    assert np.array_equal(torch.cat(out).cpu().numpy(), whole.cpu().numpy())
