"""The ISO-correct switches PINNED by an independent decoder (VERDICT r05 missing #2).

tests/golden/iso_*.npz hold what FFmpeg's mpegaudiodec (the copy inside the build container's kaleido / Chromium,
tools/ffmpeg_ref.py, tools/make_iso_golden.py) decodes twelve conforming packer streams to.  Here, without a GPU:
  * the packer still makes the bytes FFmpeg was given (SHA-256 in the fixture),
  * the oracle with PDMP3_ISO_ALL is FFmpeg's output within TOL_F32_LSB = 2 LSB (measured: 1.11 max, 0.41 rms; FFmpeg's
    decoder is the fixed-point one and rounds to int16 itself), on all twelve,
  * with any one switch off it is NOT (tens to thousands of LSB on the stream that exercises the switch): the fixtures
    discriminate, each bit is pinned -- PDMP3_ISO_IS_SHORT excepted, which PDMP3_ISO_IS_BOUND subsumes,
  * the product's host stage (records, host Huffman) + the kernels' own source as their host build give the same.
The GPU half is tests/test_gpu_iso.py::test_gpu_iso_pin."""
import hashlib
import json
import os

import numpy as np
import pytest

import iso_streams
from pdmp3_amd.packer import packer
from test_pipeline_emul import emul_decode

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ISO_ALL = 0x3f


def load_fixture(name):
    """-> (mp3 bytes, FFmpeg's int16 PCM [samples][channels], kwargs); asserts the packer reproduces the stream"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(str(z["kwargs"]))
    kw["block_pct"] = tuple(kw["block_pct"])
    kw["gain"] = tuple(kw["gain"])
    assert kw == {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in iso_streams.STREAMS[name].items()}, \
        "tests/iso_streams.py changed: run tools/make_iso_golden.py in the build container"
    mp3 = packer.generate(**kw)
    assert hashlib.sha256(mp3).hexdigest() == str(z["sha256"]), "the packer no longer makes the stream FFmpeg decoded (%s)" % name
    assert int(z["offset"]) == 0
    return mp3, z["pcm"], kw


def ffmpeg_error(ours, theirs):
    """ours: float [samples][channels] in FFmpeg's units (full scale 32768); theirs: its int16 -> (max, rms) over the
    samples both have (the reference's driver never decodes the last frame: SURVEY H10)"""
    m = min(ours.shape[0], theirs.shape[0])
    assert m >= (iso_streams.N_FRAMES - 4) * 1152
    e = np.abs(ours[:m].astype(np.float64) - theirs[:m])
    return float(e.max()), float(np.sqrt((e ** 2).mean()))


def oracle_pcm(oracle, mp3, iso, nch):
    """binary32 PCM of the oracle (P:2028's `sum`) in FFmpeg's units, [samples][channels]"""
    _, sp, sd = oracle.decode_buffer_like_cli_iso(mp3, iso, tap_frames=4096)
    _, f32 = oracle.decode_f32(sp, sd)
    n = sp.shape[0]
    x = f32.reshape(n * 1152, 2) if nch == 2 else f32[:, :1152].reshape(-1, 1)
    return np.clip(x * 32768.0, -32768.0, 32767.0)


@pytest.mark.parametrize("name", list(iso_streams.STREAMS))
def test_oracle_iso_all_is_ffmpeg(oracle, name):
    mp3, theirs, kw = load_fixture(name)
    mx, rms = ffmpeg_error(oracle_pcm(oracle, mp3, ISO_ALL, iso_streams.nch_of(kw)), theirs)
    assert mx <= iso_streams.TOL_F32_LSB and rms <= iso_streams.RMS_LSB, "%s: max %.2f LSB, rms %.3f against FFmpeg" % (name, mx, rms)


# switch -> (stream that exercises it, the least it must cost to leave it off, in LSB; measured: profiles/r06_iso_pin.json)
SWITCH_STREAM = {
    0x01: ("iso_ms_is_441", 100.0),          # TABLE33   583
    0x02: ("iso_ms_is_320", 1000.0),         # MS_BOUND  9702
    0x08: ("iso_ms_is_mixed_441", 50.0),     # SF21      141
    0x10: ("iso_ms_short_320k", 1000.0),     # SF12      11113
    0x20: ("iso_is_short_441", 500.0),       # IS_BOUND  1807
}


@pytest.mark.parametrize("bit", list(SWITCH_STREAM))
def test_every_switch_is_pinned(oracle, bit):
    name, least = SWITCH_STREAM[bit]
    mp3, theirs, kw = load_fixture(name)
    mx, _ = ffmpeg_error(oracle_pcm(oracle, mp3, ISO_ALL & ~bit, iso_streams.nch_of(kw)), theirs)
    assert mx >= least, "leaving switch %#x off costs only %.1f LSB on %s: the fixture does not pin it" % (bit, mx, name)


def test_the_reference_itself_is_not_iso_here(oracle):
    """mask 0 = the reference's behaviour, on a stream every conforming decoder agrees about: far from FFmpeg (that is
    SURVEY H1-H5, not a defect of the oracle: it equals oracle/_ref bit for bit with mask 0, tests/test_oracle.py)"""
    mp3, theirs, kw = load_fixture("iso_ms_is_441")
    mx, _ = ffmpeg_error(oracle_pcm(oracle, mp3, 0, 2), theirs)
    assert mx > 1000.0


@pytest.mark.parametrize("name", ["iso_stereo_441", "iso_ms_441", "iso_mono_320", "iso_is_short_441", "iso_ms_is_480", "iso_ms_is_mixed_441"])
def test_host_stage_and_kernel_source_are_ffmpeg(emul, name):
    """the PRODUCT's host stage (header, side info, reservoir, host Huffman, record builder: libpdmp3.so with
    pdmp3_amd_set_quirks(PDMP3_ISO_ALL)) and the kernels' source as its host build (tests/host_emul), int16 PCM"""
    from pdmp3_amd import api
    mp3, theirs, kw = load_fixture(name)
    sp, sd = api.parse_like_cli(mp3, 4096, ISO_ALL)
    pcm = emul_decode(emul, sp, sd, 0)                                  # [frames][2304] int16
    nch = iso_streams.nch_of(kw)
    x = pcm.reshape(-1, 2) if nch == 2 else pcm[:, :1152].reshape(-1, 1)
    # the reference's int16 is trunc(sum * 32767) (P:2028); FFmpeg's is round(sum * 32768)
    mx, rms = ffmpeg_error(x.astype(np.float64) * (32768.0 / 32767.0), theirs)
    assert mx <= iso_streams.TOL_S16_LSB, "%s: max %.2f LSB against FFmpeg" % (name, mx)


def load_clip_fixture():
    """the one stream here that the packer did not make: a real encoder's clip (tests/golden/clip_invalid_keypress.mp3, MathJax's
    accessibility click: 44.1 kHz stereo 64 kbps behind an ID3 tag, with an Info frame) -> (bytes, FFmpeg's int16 [samples][2], the
    index of FFmpeg's first sample in ours: Chromium trims the Info frame and the encoder delay, 1152 + 1105)"""
    z = np.load(os.path.join(GOLDEN, "iso_clip_real.npz"))
    mp3 = open(os.path.join(GOLDEN, "clip_invalid_keypress.mp3"), "rb").read()
    assert hashlib.sha256(mp3).hexdigest() == str(z["sha256"])
    return mp3, z["pcm"], int(z["offset"])


def clip_error(ours, theirs, off):
    m = min(ours.shape[0] - off, theirs.shape[0])
    assert m >= 17 * 1152
    e = np.abs(ours[off:off + m].astype(np.float64) - theirs[:m])
    return float(e.max()), float(np.sqrt((e ** 2).mean()))


def test_a_real_encoders_stream_is_ffmpeg_too(oracle, emul):
    """the packer's idea of a conforming stream is not what pins the decoder alone: a real encoder's clip through the oracle with
    PDMP3_ISO_ALL is FFmpeg's output within the same bars (measured 1.07 LSB max, 0.30 rms), and so are the product's host stage +
    the kernels' host build"""
    from pdmp3_amd import api
    mp3, theirs, off = load_clip_fixture()
    assert off == 2257
    mx, rms = clip_error(oracle_pcm(oracle, mp3, ISO_ALL, 2), theirs, off)
    assert mx <= iso_streams.TOL_F32_LSB and rms <= iso_streams.RMS_LSB, "max %.2f LSB, rms %.3f against FFmpeg" % (mx, rms)
    sp, sd = api.parse_like_cli(mp3, 4096, ISO_ALL)
    pcm = emul_decode(emul, sp, sd, 0).reshape(-1, 2)
    mx16, _ = clip_error(pcm.astype(np.float64) * (32768.0 / 32767.0), theirs, off)
    assert mx16 <= iso_streams.TOL_S16_LSB, "max %.2f LSB against FFmpeg" % mx16
