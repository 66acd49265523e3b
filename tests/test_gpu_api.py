"""The libmpg123-style API end to end on the GPU box: host Huffman stage ->
pinned batches -> hipMemcpyAsync -> HIP transforms -> PCM, against the oracle
(bit-exact restatement of the reference) on the same bytes.  Covers BASELINE
configs[0] (C1: one 44.1 kHz stereo 128 kbps CBR file through pdmp3_decode)
and a short form of configs[2] (C3: 320 kbps CBR feed/read streaming)."""
import hashlib
import os

import numpy as np
import pytest

from pdmp3_amd.packer import packer
from util import assert_pcm_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _as16(b):
    return np.frombuffer(b, dtype=np.int16)


def test_clip_through_cli_loop(oracle):
    from pdmp3_amd import api
    mp3 = open(os.path.join(GOLD, "clip_invalid_keypress.mp3"), "rb").read()
    got = api.decode_like_cli(mp3)
    want = oracle.decode_buffer_like_cli(mp3)
    assert hashlib.md5(want).hexdigest() == "691b76164c105c1f1edc5f2fe7bb7c8f"
    assert len(got) == len(want) == 92160
    assert_pcm_close(_as16(got), _as16(want), 1, "clip")


def test_c1_pdmp3_decode_128k(oracle):
    """configs[0]: single 44.1 kHz stereo 128 kbps CBR stream through pdmp3_decode()"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=300, seed=101, sfreq=0, mode=1, mode_ext=2, bitrate_index=9)
    d = api.Decoder()
    rc, _ = d.decode(mp3[:4096], 0)                    # probe: header only
    assert rc == api.PDMP3_NEW_FORMAT
    assert d.getformat()[1:3] == (44100, 2)
    out, pos = [], 4096
    while True:
        chunk = mp3[pos:pos + 3000]
        pos += len(chunk)
        rc, pcm = d.decode(chunk, 16384) if chunk else d.read(16384)
        out.append(pcm)
        if not chunk and rc != api.PDMP3_OK:
            break
    d.close()
    got = b"".join(out)
    # same call sequence on the oracle's API restatement
    import ctypes as C
    L = oracle.lib
    L.orc_stream_new.restype = C.c_void_p
    s = C.c_void_p(L.orc_stream_new())
    L.orc_stream_open_feed(s)
    done = C.c_size_t(0)
    buf = (C.c_ubyte * 16384)()
    L.orc_stream_decode(s, mp3[:4096], 4096, None, 0, C.byref(done))
    lr, lc, le = C.c_long(), C.c_int(), C.c_int()
    L.orc_stream_getformat(s, C.byref(lr), C.byref(lc), C.byref(le))
    want, pos = [], 4096
    while True:
        chunk = mp3[pos:pos + 3000]
        pos += len(chunk)
        if chunk:
            rc = L.orc_stream_decode(s, chunk, len(chunk), buf, 16384, C.byref(done))
        else:
            rc = L.orc_stream_read(s, buf, 16384, C.byref(done))
        want.append(bytes(buf[:done.value]))
        if not chunk and rc != 0:
            break
    want = b"".join(want)
    # pdmp3_decode silently drops input beyond the ring's free space (H16): fewer than 300 frames come out, identically
    assert len(got) == len(want) and len(got) > 100 * 4608
    assert_pcm_close(_as16(got), _as16(want), 1, "C1")


@pytest.mark.parametrize("kw", [
    dict(n_frames=400, seed=102, sfreq=0, mode=1, mode_ext=2, bitrate_index=14),            # C3 short form
    dict(n_frames=200, seed=103, sfreq=2, mode=3, bitrate_index=7),
    dict(n_frames=200, seed=104, sfreq=1, mode=0, mode_ext=0, vbr=True, crc=True, table33_pct=10),
    dict(n_frames=150, seed=105, block_pct=(10, 10, 70, 10), bitrate_index=12),
])
def test_feed_read_streaming(oracle, kw):
    from pdmp3_amd import api
    mp3 = packer.generate(**kw)
    got = api.decode_like_cli(mp3)
    want = oracle.decode_buffer_like_cli(mp3)
    assert len(got) == len(want)
    assert_pcm_close(_as16(got), _as16(want), 1, str(kw))


def test_two_handles_do_not_share_state(oracle):
    """per-handle synthesis state (the reference's is process-global); a
    handle re-opened with pdmp3_open_feed keeps its parse state (stale count1 /
    scalefactors, H4-H6) exactly like the reference -- replayed on the oracle."""
    from oracle.oracle import OracleStream
    from pdmp3_amd import api
    a = packer.generate(n_frames=60, seed=201)
    b = packer.generate(n_frames=60, seed=202, sfreq=1)
    da, db = api.Decoder(), api.Decoder()
    oa, ob = OracleStream(oracle), OracleStream(oracle)
    pa = api.decode_like_cli(a, da)
    pb = api.decode_like_cli(b, db)                   # interleaved use of a second handle
    pa2 = api.decode_like_cli(b, da)                  # handle a re-opened on another stream
    pb2 = api.decode_like_cli(a, db)
    da.close(); db.close()
    for got, want, what in ((pa, oa.decode_like_cli(a), "a"), (pb, ob.decode_like_cli(b), "b"),
                            (pa2, oa.decode_like_cli(b), "a reopened"), (pb2, ob.decode_like_cli(a), "b reopened")):
        assert len(got) == len(want)
        assert_pcm_close(_as16(got), _as16(want), 1, what)
    oa.close(); ob.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_call_sequences_on_the_gpu(oracle, seed):
    """the read-ahead inside pdmp3_read is invisible: random feed / read / decode / getformat / open_feed sequences,
    every return code, byte count and PCM byte (+-1 LSB) against the oracle's restatement of the reference's API --
    including the sequences that make the library take frames back (a feed that fills the ring exactly, open_feed)"""
    import stream_replay
    from oracle.oracle import OracleStream
    from pdmp3_amd import api
    for name, mp3 in stream_replay.streams():
        dec = api.Decoder()
        orc = OracleStream(oracle)
        try:
            stream_replay.replay(1000 * seed + len(name), mp3, dec, orc, compare_pcm=True)
        finally:
            dec.close()
            orc.close()


def test_stream_loop_in_c_matches_the_cli_loop(oracle):
    """pdmp3_amd_stream_loop (the driver's loop in C: what bench.py times as streaming_api), at the reference's cadence
    and with eager feeds / other buffer sizes: same bytes as the reference's loop on the oracle"""
    from oracle.oracle import OracleStream
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=500, seed=77, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    o = OracleStream(oracle)
    want = _as16(o.decode_like_cli(mp3))
    o.close()
    n, got = api.stream_loop(mp3)
    assert n == want.nbytes
    assert_pcm_close(got, want, 1, "reference cadence")
    for feed, read, eager in ((4096, 16384, True), (1000, 4608, True), (15000, 65536, True), (8192, 100000, False)):
        n2, got2 = api.stream_loop(mp3, feed, read, eager)
        # other cadences deliver the same frames except for the tail rule (H10: a frame is attempted only with 1152
        # bytes buffered), which depends on how the end of the file is fed
        k = min(got2.size, want.size)
        assert k >= want.size - 2 * 2304 * 2 and np.array_equal(got2[:k], got[:k]), (feed, read, eager)


def test_float_encoding_through_the_api(oracle):
    """pdmp3_amd_set_encoding(PDMP3_ENC_FLOAT_32): pdmp3_read delivers the unscaled synthesis sums as float (8 bytes per
    stereo sample-frame); against the oracle's floats for the records the oracle parses from the same bytes; switching
    back mid-stream gives the int16 stream again"""
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=120, seed=91, sfreq=0, mode=1, mode_ext=2, bitrate_index=12, block_pct=(55, 15, 15, 15))
    pcm16, sp, sd = oracle.decode_buffer_like_cli(mp3, tap_frames=200)
    _, want = oracle.decode_f32(sp, sd)
    want = want.reshape(-1)
    d = api.Decoder()
    assert d.set_encoding(0x123) == api.PDMP3_ERR
    assert d.set_encoding(api.PDMP3_ENC_FLOAT_32) == 0
    out, pos, half = [], 0, False
    while True:
        rc, pcm = d.read(10000)                           # (not a multiple of the frame size: the cursor is exercised)
        if rc == api.PDMP3_ERR:
            break
        out.append(pcm)
        if rc == api.PDMP3_NEED_MORE:
            chunk = mp3[pos:pos + 4096]
            if not chunk:
                break
            d.feed(chunk)
            pos += len(chunk)
    assert d.getformat()[3] == api.PDMP3_ENC_FLOAT_32
    d.close()
    got = np.frombuffer(b"".join(out), dtype=np.float32)
    assert got.size == want.size == len(pcm16) // 2
    assert float(np.abs(got - want).max()) <= 1e-5 * max(1.0, float(np.abs(want).max()))
