"""Stream-level BASELINE configs on the GPU box:

  configs[2] (C3)  feed/read streaming of a long 44.1 kHz stereo 320 kbps CBR stream, host Huffman +
                   GPU transforms: the full hour, 137 813 frames, ~144 MB (PDMP3_SHORT_C3=1 runs 20 000 frames)
  configs[3] (C4)  mixed corpus: mono / stereo / joint-MS x 32 / 44.1 / 48 kHz x CBR+VBR x long /
                   start-short-stop / mixed blocks, files dealt to ranks largest-first
  CLI             pdmp3_cli (the reference's main.c contract) writing <file>.raw

Oracle = bit-exact restatement of the reference, on the same bytes; bar +-1 LSB.
"""
import hashlib
import os
import subprocess
import time

import numpy as np
import pytest

from pdmp3_amd.packer import packer
from util import assert_pcm_close, level, oracle_decode_many

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _as16(b):
    return np.frombuffer(b, dtype=np.int16)


def test_c3_long_stream(oracle):
    """BASELINE configs[2] at full size: one hour (137 813 frames) of 44.1 kHz joint stereo 320 kbps CBR through the
    drop-in API's feed / read loop (PDMP3_SHORT_C3=1: 20 000 frames)"""
    from pdmp3_amd import api
    n = 20000 if os.environ.get("PDMP3_SHORT_C3") else 137813
    mp3 = packer.generate(n_frames=n, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    assert abs(len(mp3) / n - 1044.9) < 0.2                     # 320 kbps CBR with ISO padding
    t0 = time.time()
    nbytes, got16 = api.stream_loop(mp3)                        # pdmp3_feed / pdmp3_read at the reference driver's cadence, in C
    dt = time.time() - t0
    got = got16.tobytes()
    head = api.decode_like_cli(mp3[:3000 * 1045])               # the same loop driven from Python
    assert head[:2900 * 4608] == got[:2900 * 4608]
    want = oracle.decode_buffer_like_cli(mp3)
    assert len(got) == len(want) >= (n - 3) * 4608
    dmax, ndiff = assert_pcm_close(_as16(got), _as16(want), 1, "C3")
    fps = (len(got) / 4608) / dt
    print("C3: %d frames in %.2f s = %.0f frames/s = %.0fx real time (feed/read loop in C); "
          "max diff %d LSB in %.3f%% of samples" % (len(got) // 4608, dt, fps, fps / 38.28125, dmax, 100.0 * ndiff / (len(got) // 2)))
    assert fps / 38.28125 >= 50, "north_star: >= 50x real time"
    # the same bytes through the whole-stream decoder (device Huffman, then host Huffman): identical PCM, bit for bit
    for host_huffman in (False, True):
        b = api.BulkDecoder(threads=4, host_huffman=host_huffman)
        try:
            t0 = time.time()
            bulk = b.decode(mp3)
            dt = time.time() - t0
        finally:
            b.close()
        assert bulk.nbytes == len(got) and np.array_equal(bulk, _as16(got)), "bulk (host_huffman=%s) != streaming API" % host_huffman
        print("C3 bulk (%s Huffman): %.3f s = %.0fx real time" % ("host" if host_huffman else "device", dt, (len(got) / 4608) / dt / 38.28125))


def test_c4_mixed_corpus(oracle):
    from pdmp3_amd import api
    from pdmp3_amd.sharding import assign_files
    files = []
    seed = 400
    for mode, mext in ((3, 0), (0, 0), (1, 2)):
        for sfreq in (0, 1, 2):
            for vbr in (False, True):
                for blocks in ((100, 0, 0, 0), (40, 20, 20, 20), (20, 10, 60, 10)):
                    seed += 1
                    hi = 13 if sfreq == 2 else 14                # 32 kHz <= 256 kbps (H10)
                    files.append(packer.generate(n_frames=40 + seed % 50, seed=seed, sfreq=sfreq, mode=mode, mode_ext=mext,
                                                 vbr=vbr, vbr_lo=4, vbr_hi=hi, bitrate_index=min(12, hi), block_pct=blocks,
                                                 mixed_pct=50 if blocks[2] else 0))
    assert len(files) == 54
    plan = assign_files([len(f) for f in files], 8)
    assert sorted(sum(plan, [])) == list(range(len(files)))
    dec = [api.Decoder() for _ in range(2)]                     # two live handles, interleaved
    worst = 0
    for r, idxs in enumerate(plan):
        for i in idxs:
            d = api.Decoder()                                   # fresh handle per file = fresh parse state, like the oracle
            got = api.decode_like_cli(files[i], d)
            d.close()
            want = oracle.decode_buffer_like_cli(files[i])
            assert len(got) == len(want)
            dmax, _ = assert_pcm_close(_as16(got), _as16(want), 1, "file %d" % i)
            worst = max(worst, dmax)
    for d in dec:
        d.close()
    assert worst <= 1


LOUD = (145, 160)           # packer global_gain range that puts the PCM at a realistic level: median |sample| 70-250 LSB,
                            # 99th percentile at full scale, 1-4 % of the samples clipped (default 110..150: median 1-9 LSB)


def _c4_files(n_base=40, n_mod=50, hi_32k=13, seed0=400):
    """hi_32k: top bitrate index at 32 kHz.  13 = 256 kbps = 1152-byte frames, the H10 limit -- long streams of
    those drive the reference into replaying its input ring (include/pdmp3_bulk.h PDMP3_BULK_REPLAY), so the
    full-size corpus stops at 12.  Every other file is LOUD (the +-1 LSB bar on a signal that is not nearly silent)."""
    files, seed = [], seed0
    for mode, mext in ((3, 0), (0, 0), (1, 2)):
        for sfreq in (0, 1, 2):
            for vbr in (False, True):
                for blocks in ((100, 0, 0, 0), (40, 20, 20, 20), (20, 10, 60, 10)):
                    seed += 1
                    hi = hi_32k if sfreq == 2 else 14
                    files.append(packer.generate(n_frames=n_base + seed % n_mod, seed=seed, sfreq=sfreq, mode=mode, mode_ext=mext,
                                                 vbr=vbr, vbr_lo=4, vbr_hi=hi, bitrate_index=min(12, hi), block_pct=blocks,
                                                 mixed_pct=50 if blocks[2] else 0, gain=LOUD if seed % 2 else (110, 150)))
    return files


def test_c4_bulk_decoders_in_parallel(oracle):
    """BASELINE configs[3] at FULL size (>= 64 files x >= 4096 frames; 64 files, 264 k frames): {mono, stereo,
    joint-MS} x {32, 44.1, 48 kHz} x {CBR, VBR} x {long, start/short/stop, mixed}, half of the files LOUD, through the
    bulk decoder (device Huffman), six decoders live at once on their own host threads and HIP streams, files dealt
    largest-first like ranks would take them; every sample against the oracle (a thread pool: 36 s of reference-speed
    work on one core), +-1 LSB.  PDMP3_SHORT_C4=1: 54 files x 41-89 frames."""
    import threading
    from pdmp3_amd import api
    from pdmp3_amd.sharding import assign_files
    short = bool(os.environ.get("PDMP3_SHORT_C4"))
    files = _c4_files() if short else _c4_files(4096, 64, 12) + _c4_files(4096, 64, 12, seed0=900)[::5][:10]
    assert len(files) >= (54 if short else 64)
    jobs = 6
    plan = assign_files([len(f) for f in files], jobs)
    assert sorted(sum(plan, [])) == list(range(len(files)))
    got = [None] * len(files)

    def work(j):
        b = api.BulkDecoder(threads=2, window_frames=16 if short else 512)
        try:
            for i in plan[j]:
                got[i] = b.decode(files[i])
        finally:
            b.close()
    t0 = time.time()
    ts = [threading.Thread(target=work, args=(j,)) for j in range(jobs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.time() - t0
    t0 = time.time()
    want = oracle_decode_many(oracle, files)
    dt_cpu = time.time() - t0
    frames = worst = ndiff = 0
    loud_med = []
    for i in range(len(files)):
        nch = 1 if (files[i][3] >> 6) == 3 else 2
        assert got[i] is not None and got[i].shape == want[i].shape, i
        dmax, nd = assert_pcm_close(got[i], want[i], 1, "file %d" % i)
        worst, ndiff = max(worst, dmax), ndiff + nd
        frames += got[i].size // 1152 // nch
        loud_med.append(level(want[i])[0])
        want[i] = got[i] = None
    assert short or frames >= 64 * 4096
    print("C4 bulk: %d files, %d frames, %d decoders in parallel: %.3f s = %.0f frames/s; oracle %.1f s; max diff %d LSB (%d "
          "samples); median |PCM| per file %d..%d LSB" % (len(files), frames, jobs, dt, frames / dt, dt_cpu, worst, ndiff,
                                                         min(loud_med), max(loud_med)))


@pytest.mark.parametrize("mode,mext", [(3, 0), (2, 0), (0, 0), (1, 2), (1, 0)], ids=["mono", "dual", "stereo", "joint_ms", "joint_off"])
@pytest.mark.parametrize("sfreq", [0, 1, 2], ids=["441", "480", "320"])
def test_loud_streams_every_mode(oracle, mode, mext, sfreq):
    """+-1 LSB on streams that are neither silent nor clipped, per mode x rate: 600 frames with
    long / start / short / stop / MIXED blocks at the LOUD gain, through the streaming API (host Huffman) and the
    whole-stream decoder (device Huffman) -- the level is asserted, not assumed"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=600, seed=0x10D + 16 * mode + sfreq, sfreq=sfreq, mode=mode, mode_ext=mext,
                          bitrate_index=12 if sfreq == 2 else 13, block_pct=(40, 15, 30, 15), mixed_pct=50, gain=LOUD)
    want = np.frombuffer(oracle.decode_buffer_like_cli(mp3), dtype=np.int16)
    med, p99, clipped = level(want)
    assert med >= 40 and p99 >= 8000 and clipped < 0.06, (med, p99, clipped)
    got = _as16(api.decode_like_cli(mp3))
    assert got.shape == want.shape
    assert_pcm_close(got, want, 1, "streaming API")
    b = api.BulkDecoder(threads=2, window_frames=64)
    try:
        bulk = b.decode(mp3)
    finally:
        b.close()
    assert np.array_equal(bulk, got), "bulk != streaming API"


def test_cli_writes_raw(oracle, tmp_path):
    mp3 = packer.generate(n_frames=200, seed=0xC1, sfreq=0, mode=1, mode_ext=2, bitrate_index=9)
    path = tmp_path / "c1_128k.mp3"
    path.write_bytes(mp3)
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    subprocess.check_call([cli, str(path)], timeout=120)
    got = (tmp_path / "c1_128k.mp3.raw").read_bytes()
    want = oracle.decode_buffer_like_cli(mp3)
    assert len(got) == len(want)
    assert_pcm_close(_as16(got), _as16(want), 1, "CLI")


@pytest.mark.parametrize("streaming", ["0", "1"])
def test_cli_writes_wav(oracle, tmp_path, streaming):
    """PDMP3_CLI_WAV=1: the same samples as <file>.wav, header with the stream's rate and channel count (the
    reference's only sink is the raw writer, pdmp3.c:2236-2257); and pdmp3_amd_write_wav for buffers, int16 and float"""
    import ctypes as C
    import wave
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=90, seed=0xC7, sfreq=1, mode=3, bitrate_index=8)       # 48 kHz mono
    path = tmp_path / "m.mp3"
    path.write_bytes(mp3)
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    subprocess.check_call([cli, str(path)], timeout=120, env=dict(os.environ, PDMP3_CLI_WAV="1", PDMP3_CLI_STREAMING=streaming))
    want = oracle.decode_buffer_like_cli(mp3)
    with wave.open(str(tmp_path / "m.mp3.wav"), "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 48000)
        assert w.getnframes() == len(want) // 2
        got = w.readframes(w.getnframes())
    assert_pcm_close(_as16(got), _as16(want), 1, "wav")
    lib = api.load_library()
    lib.pdmp3_amd_write_wav.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_long, C.c_int, C.c_int]
    out = str(tmp_path / "buf.wav").encode()
    pcm = np.frombuffer(want, dtype=np.int16)
    assert lib.pdmp3_amd_write_wav(out, pcm.ctypes.data_as(C.c_void_p), pcm.nbytes, 48000, 1, 0) == 0
    with wave.open(out.decode(), "rb") as w:
        assert w.readframes(w.getnframes()) == want
    f32 = (pcm.astype(np.float32) / 32767.0)
    assert lib.pdmp3_amd_write_wav(out, f32.ctypes.data_as(C.c_void_p), f32.nbytes, 48000, 1, 1) == 0
    raw = open(out.decode(), "rb").read()
    assert raw[20:22] == b"\x03\x00" and raw[34:36] == b"\x20\x00" and raw[44:] == f32.tobytes()


@pytest.mark.parametrize("streaming", ["0", "1"])
def test_cli_names_its_output_after_the_first_file_even_if_that_one_is_empty(oracle, tmp_path, streaming):
    """the reference's driver calls its writer after EVERY pdmp3_read, the first NEED_MORE with nothing decoded included
    (pdmp3.c:2565-2566, 2239-2251): the .raw is named after the FIRST file even if that file yields no PCM, and a file
    without a decodable frame still leaves an (empty) .raw"""
    good = packer.generate(n_frames=40, seed=0xE1, sfreq=0, mode=1, mode_ext=2, bitrate_index=9)
    bad = tmp_path / "bad.mp3"
    bad.write_bytes(good[:700])                       # less than one buffered frame's worth: nothing is decoded (H10)
    ok = tmp_path / "good.mp3"
    ok.write_bytes(good)
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    env = dict(os.environ, PDMP3_CLI_STREAMING=streaming)
    subprocess.check_call([cli, str(bad), str(ok)], timeout=120, env=env)
    assert not (tmp_path / "good.mp3.raw").exists()
    got = (tmp_path / "bad.mp3.raw").read_bytes()
    from oracle.oracle import OracleStream
    o = OracleStream(oracle)
    want = o.decode_like_cli(good[:700]) + o.decode_like_cli(good)
    o.close()
    assert len(got) == len(want)
    assert_pcm_close(_as16(got), _as16(want), 1, "bad + good")
    (tmp_path / "bad.mp3.raw").unlink()
    subprocess.check_call([cli, str(bad)], timeout=120, env=env)
    assert (tmp_path / "bad.mp3.raw").read_bytes() == o_empty(oracle, good[:700])


def o_empty(oracle, data):
    from oracle.oracle import OracleStream
    o = OracleStream(oracle)
    try:
        return o.decode_like_cli(data)
    finally:
        o.close()


def test_cli_several_files_one_handle(oracle, tmp_path):
    """pdmp3() decodes all its files with ONE handle into "<first name>.raw": parse state left by a file shows in
    the next.  The whole-stream path the CLI takes for regular files carries it the same way;
    PDMP3_CLI_STREAMING=1 (the reference's own feed/read loop) must give the identical bytes."""
    from oracle.oracle import OracleStream
    specs = [dict(n_frames=60, seed=0xD1, mode=1, mode_ext=2, bitrate_index=11, block_pct=(10, 10, 70, 10), mixed_pct=50),
             dict(n_frames=45, seed=0xD2, sfreq=1, mode=3, bitrate_index=8, block_pct=(30, 20, 30, 20)),
             dict(n_frames=70, seed=0xD3, sfreq=2, mode=0, mode_ext=0, vbr=True, vbr_lo=3, vbr_hi=12, block_pct=(20, 10, 60, 10))]
    files = [packer.generate(**s) for s in specs]
    paths = []
    for i, f in enumerate(files):
        p = tmp_path / ("f%d.mp3" % i)
        p.write_bytes(f)
        paths.append(str(p))
    o = OracleStream(oracle)
    want = b"".join(o.decode_like_cli(f) for f in files)
    o.close()
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    raw = tmp_path / "f0.mp3.raw"
    subprocess.check_call([cli] + paths, timeout=120)
    got = raw.read_bytes()
    raw.unlink()
    subprocess.check_call([cli] + paths, timeout=120, env=dict(os.environ, PDMP3_CLI_STREAMING="1"))
    got_loop = raw.read_bytes()
    assert len(got) == len(want) == len(got_loop)
    assert got == got_loop
    assert_pcm_close(_as16(got), _as16(want), 1, "3 files, one handle")
