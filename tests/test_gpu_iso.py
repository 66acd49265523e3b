"""The ISO-correct switches on the GPU: the HIP engine on records that carry PDMP3_GC_ISO_*, the
whole-stream decoder (device Huffman and host Huffman), the streaming API and the CLI with pdmp3_amd_set_quirks /
$PDMP3_CLI_ISO -- against the oracle's restatement of the same switches (+-1 LSB), and, since round 6, against what an
INDEPENDENT ISO decoder (FFmpeg) made of the same bytes: test_gpu_iso_pin, fixtures tests/golden/iso_*.npz."""
import os
import subprocess

import numpy as np
import pytest

import corpus
from test_gpu_parity import gpu_decode
import iso_streams
from test_iso_pin import ffmpeg_error, load_fixture
from test_iso_switches import _streams, ISO_TABLE33, ISO_MS_BOUND, ISO_IS_SHORT, ISO_SF21, ISO_SF12, ISO_IS_BOUND, ISO_ALL
from util import assert_pcm_close, nch_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(corpus.ISO_CASES))
def test_gpu_engine_on_iso_records(engine, oracle, name):
    sp, sd = corpus.case(name)
    want, ws = oracle.decode(sp, sd, stages=True)
    got, gs = gpu_decode(engine, sp, sd, stages=True)
    nch = nch_of(sd)
    for k in range(3):
        assert np.array_equal(ws[:, :, :nch, k].view(np.uint32), gs[:, :, :nch, k].view(np.uint32)), "stage %d" % k
    assert_pcm_close(got, want, 1, name)
    for chunk in (0, 1, 3):   # granule kernel, chunks with halos (the persistent kernel: tests/ring_variant_checks.py)
        assert np.array_equal(gpu_decode(engine, sp, sd, chunk=chunk), got), chunk


@pytest.mark.parametrize("name", list(iso_streams.STREAMS))
def test_gpu_iso_pin(name):
    """PDMP3_ISO_ALL through every product path -- streaming API (float32 and int16 PCM), whole-stream decoder with the
    Huffman stage on the device and on the host -- against FFmpeg's decode of the same bytes: <= 2 LSB for float PCM,
    <= 3 LSB for int16 (truncated toward zero, P:2028), literal bars of tests/iso_streams.py"""
    from pdmp3_amd import api
    mp3, theirs, kw = load_fixture(name)
    nch = iso_streams.nch_of(kw)
    outs = {}
    for enc, dt in ((api.PDMP3_ENC_FLOAT_32, np.float32), (api.PDMP3_ENC_SIGNED_16, np.int16)):
        d = api.Decoder()                                  # (a fresh handle each: parse state survives pdmp3_open_feed)
        try:
            d.set_quirks(ISO_ALL)
            assert d.set_encoding(enc) == 0
            outs[enc] = np.frombuffer(api.decode_like_cli(mp3, d), dtype=dt).reshape(-1, nch)
        finally:
            d.close()
    f32, s16 = outs[api.PDMP3_ENC_FLOAT_32], outs[api.PDMP3_ENC_SIGNED_16]
    mx, rms = ffmpeg_error(np.clip(f32.astype(np.float64) * 32768.0, -32768.0, 32767.0), theirs)
    assert mx <= iso_streams.TOL_F32_LSB and rms <= iso_streams.RMS_LSB, "%s float PCM: max %.2f LSB, rms %.3f against FFmpeg" % (name, mx, rms)
    mx16, _ = ffmpeg_error(s16.astype(np.float64) * (32768.0 / 32767.0), theirs)
    assert mx16 <= iso_streams.TOL_S16_LSB, "%s int16 PCM: max %.2f LSB against FFmpeg" % (name, mx16)
    for host_huffman in (False, True):
        b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
        try:
            b.set_quirks(ISO_ALL)
            got = b.decode(mp3).reshape(-1, nch)
        finally:
            b.close()
        assert np.array_equal(got, s16), "whole-stream decoder (host_huffman=%s) != streaming API" % host_huffman


def test_gpu_iso_pin_real_encoder_clip():
    """the same on the one stream the packer did not make (tests/test_iso_pin.py load_clip_fixture): streaming API in float and
    int16, whole-stream decoder with device and host Huffman, against FFmpeg's decode of the clip"""
    from pdmp3_amd import api
    from test_iso_pin import clip_error, load_clip_fixture
    mp3, theirs, off = load_clip_fixture()
    outs = {}
    for enc, dt in ((api.PDMP3_ENC_FLOAT_32, np.float32), (api.PDMP3_ENC_SIGNED_16, np.int16)):
        d = api.Decoder()
        try:
            d.set_quirks(ISO_ALL)
            assert d.set_encoding(enc) == 0
            outs[enc] = np.frombuffer(api.decode_like_cli(mp3, d), dtype=dt).reshape(-1, 2)
        finally:
            d.close()
    f32, s16 = outs[api.PDMP3_ENC_FLOAT_32], outs[api.PDMP3_ENC_SIGNED_16]
    mx, rms = clip_error(np.clip(f32.astype(np.float64) * 32768.0, -32768.0, 32767.0), theirs, off)
    assert mx <= iso_streams.TOL_F32_LSB and rms <= iso_streams.RMS_LSB, "float PCM: max %.2f LSB, rms %.3f against FFmpeg" % (mx, rms)
    mx16, _ = clip_error(s16.astype(np.float64) * (32768.0 / 32767.0), theirs, off)
    assert mx16 <= iso_streams.TOL_S16_LSB, "int16 PCM: max %.2f LSB against FFmpeg" % mx16
    for host_huffman in (False, True):
        b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
        try:
            b.set_quirks(ISO_ALL)
            got = b.decode(mp3).reshape(-1, 2)
        finally:
            b.close()
        assert np.array_equal(got, s16), "whole-stream decoder (host_huffman=%s) != streaming API" % host_huffman


@pytest.mark.parametrize("iso", [ISO_TABLE33, ISO_MS_BOUND | ISO_IS_SHORT, ISO_SF21 | ISO_SF12, ISO_MS_BOUND | ISO_IS_BOUND, ISO_ALL])
def test_gpu_streams_with_quirks(oracle, iso):
    from pdmp3_amd import api
    for name, mp3 in _streams().items():
        want = np.frombuffer(oracle.decode_buffer_like_cli_iso(mp3, iso), dtype=np.int16)
        ref = np.frombuffer(oracle.decode_buffer_like_cli(mp3), dtype=np.int16)
        outs = []
        for host_huffman in (False, True):
            b = api.BulkDecoder(threads=2, window_frames=32, host_huffman=host_huffman)
            try:
                b.set_quirks(iso)
                outs.append(b.decode(mp3))
                b.set_quirks(0)
                back = b.decode(mp3)                   # and back to the reference's behaviour
            finally:
                b.close()
            assert back.shape == ref.shape
            assert_pcm_close(back, ref, 1, name + " mask 0 again")
        assert outs[0].shape == want.shape and np.array_equal(outs[0], outs[1]), name
        assert_pcm_close(outs[0], want, 1, "%s bulk, iso %#x" % (name, iso))
        d = api.Decoder()
        d.set_quirks(iso)
        got = np.frombuffer(api.decode_like_cli(mp3, d), dtype=np.int16)
        d.close()
        assert np.array_equal(got, outs[0]), "streaming API != whole-stream decoder"


def test_cli_iso_env(oracle, tmp_path):
    from pdmp3_amd.packer import packer
    mp3 = _streams()["joint_ms_is_t33"]
    path = tmp_path / "q.mp3"
    path.write_bytes(mp3)
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    for streaming in ("0", "1"):
        subprocess.check_call([cli, str(path)], timeout=120, env=dict(os.environ, PDMP3_CLI_ISO="0x3f", PDMP3_CLI_STREAMING=streaming))
        got = np.frombuffer((tmp_path / "q.mp3.raw").read_bytes(), dtype=np.int16)
        (tmp_path / "q.mp3.raw").unlink()
        want = np.frombuffer(oracle.decode_buffer_like_cli_iso(mp3, ISO_ALL), dtype=np.int16)
        assert got.shape == want.shape
        assert_pcm_close(got, want, 1, "CLI, PDMP3_CLI_ISO=0x3f, streaming=" + streaming)
