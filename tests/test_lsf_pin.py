"""MPEG-2 LSF / MPEG-2.5 (last third; PDMP3_ISO_LSF) without a GPU, pinned by an independent decoder.

tests/golden/lsf_*.npz hold FFmpeg's decode (tools/make_iso_golden.py, build container only) of 24 conforming packer
streams: {stereo, mono, M/S, M/S + intensity} x {22.05, 24, 16, 11.025, 12, 8 kHz}, every block type.  The reference
rejects these streams (pdmp3.c:1293) -- there is nothing of its to reproduce, so the bar is the other decoder:
  * the oracle (ORC_ISO_LSF) is within 2 LSB of FFmpeg on every fixture -- 24 kHz: 6 LSB, because FFmpeg's long band
    table has 330 where the standard has 332; with that ONE entry changed in the oracle (orc_debug_24k_330) it is 2 LSB too;
  * the product's host stage (libpdmp3.so: header, side info, the 9-bit scalefac_compress partitions, regions) builds
    records byte-identical to the oracle's, and rejects the streams without the bit, like the reference;
  * the kernels' source as its host build (the engine's LSF launch: frames regrouped in pairs, an odd last one) decodes
    those records to FFmpeg's PCM.
The GPU half is tests/test_gpu_lsf.py."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import iso_streams
from pdmp3_amd.packer import packer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ISO_LSF = 0x40


def load_lsf_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = json.loads(str(z["kwargs"]))
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()}
    assert kw == {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in iso_streams.LSF_STREAMS[name].items()}, \
        "tests/iso_streams.py changed: run tools/make_iso_golden.py in the build container"
    mp3 = packer.generate(**kw)
    assert hashlib.sha256(mp3).hexdigest() == str(z["sha256"]), "the packer no longer makes the stream FFmpeg decoded (%s)" % name
    assert int(z["rate"]) == iso_streams.lsf_rate_of(kw) and int(z["offset"]) == 0
    return mp3, z["pcm"], kw


def ffmpeg_error(ours, theirs, min_frames=25):
    m = min(ours.shape[0], theirs.shape[0])
    assert m >= min_frames * 576, (ours.shape, theirs.shape)
    e = np.abs(ours[:m].astype(np.float64) - theirs[:m])
    return float(e.max()), float(np.sqrt((e ** 2).mean()))


def lsf_pcm_of_records(f32_or_s16, n, nch):
    """[frames][2304] in the oracle's per-frame layout (an LSF frame fills the first 576 x nch) -> [samples][channels]"""
    return f32_or_s16[:, :576 * nch].reshape(n * 576, nch)


def oracle_pcm(oracle, mp3, nch):
    _, sp, sd = oracle.decode_buffer_like_cli_iso(mp3, ISO_LSF, tap_frames=4096)
    _, f32 = oracle.decode_f32(sp, sd)
    return np.clip(lsf_pcm_of_records(f32, sp.shape[0], nch) * 32768.0, -32768.0, 32767.0), sp, sd


@pytest.mark.parametrize("name", list(iso_streams.LSF_STREAMS))
def test_oracle_lsf_is_ffmpeg(oracle, name):
    mp3, theirs, kw = load_lsf_fixture(name)
    nch = iso_streams.nch_of(kw)
    ours, sp, sd = oracle_pcm(oracle, mp3, nch)
    assert (sd["lsf"][:, 0, 0] & 3 == kw["version"]).all()
    mx, rms = ffmpeg_error(ours, theirs)
    assert mx <= iso_streams.LSF_TOL_F32_LSB[name] and rms <= iso_streams.RMS_LSB, "%s: max %.2f LSB, rms %.3f against FFmpeg" % (name, mx, rms)


def test_the_24_khz_residual_is_one_table_entry(oracle):
    """FFmpeg's 24 kHz long band table starts band 18 at line 330, the standard's (ours) at 332: with the oracle's entry
    set to FFmpeg's, the one fixture that shows the difference is within the 2 LSB of all the others"""
    mp3, theirs, kw = load_lsf_fixture("lsf_24k_msis")
    oracle.lib.orc_debug_24k_330(1)
    try:
        mx330, _ = ffmpeg_error(oracle_pcm(oracle, mp3, 2)[0], theirs)
    finally:
        oracle.lib.orc_debug_24k_330(0)
    mx332, _ = ffmpeg_error(oracle_pcm(oracle, mp3, 2)[0], theirs)
    assert mx330 <= iso_streams.TOL_F32_LSB < mx332 <= iso_streams.LSF_TOL_F32_LSB["lsf_24k_msis"], (mx330, mx332)


@pytest.mark.parametrize("name", list(iso_streams.LSF_STREAMS))
def test_host_stage_builds_the_oracles_records(oracle, name):
    from pdmp3_amd import api
    mp3, _, kw = load_lsf_fixture(name)
    _, sp_o, sd_o = oracle.decode_buffer_like_cli_iso(mp3, ISO_LSF, tap_frames=4096)
    sp_h, sd_h = api.parse_like_cli(mp3, 4096, ISO_LSF)
    assert sp_h.shape[0] == sp_o.shape[0] >= 25
    assert np.array_equal(sp_h, sp_o) and np.array_equal(sd_h.view(np.uint8), sd_o.view(np.uint8)), name
    assert (sp_h[:, 1] == 0).all() and (sd_h["count1"][:, 1] == 0).all()            # an LSF frame is one granule
    if name.endswith("_msis"):
        assert (sd_h["lsf_nsfb"][:, 0, 1].sum(axis=1) > 0).all() and (sd_h["lsf_nsfb"][:, 0, 0] == 0).all()


def test_without_the_bit_lsf_is_rejected_like_the_reference_does(oracle):
    from pdmp3_amd import api
    mp3, _, _ = load_lsf_fixture("lsf_22k_stereo")
    for iso in (0, 0x3f):
        sp, sd = api.parse_like_cli(mp3, 100, iso)
        assert sp.shape[0] == 0
        assert oracle.decode_buffer_like_cli_iso(mp3, iso) == b""
    d = api.Decoder(parse_only=True)
    try:
        with pytest.raises(ValueError):
            d.set_quirks(0x80)
        d.set_quirks(0x7f)
    finally:
        d.close()


def emul_lsf(emul, sp, sd, chunk=0, f32=False):
    """tests/host_emul emul_decode_lsf_frames: the engine's LSF launch (pairs of frames + DecodeArgs::n_gran) on the host"""
    n = sp.shape[0]
    sp = np.ascontiguousarray(sp)
    sd = np.ascontiguousarray(sd)
    pcm = np.zeros(((n + 1) // 2, 2304), dtype=np.float32 if f32 else np.int16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    emul.emul_decode_lsf_frames(p(sp), p(sd), n, None, None if f32 else p(pcm), p(pcm) if f32 else None, chunk)
    return pcm


def pairs_to_samples(pcm, n, nch):
    """the engine's LSF PCM layout (include/pdmp3_hip.h pdmp3_hip_decode_lsf_frames) -> [samples][channels]"""
    return pcm[:, :1152 * nch].reshape(-1, nch)[:n * 576]


@pytest.mark.parametrize("name", [n for n in iso_streams.LSF_STREAMS if not n.endswith("_stereo")])
def test_kernel_source_decodes_lsf_records_to_ffmpegs_pcm(emul, oracle, name):
    from pdmp3_amd import api
    mp3, theirs, kw = load_lsf_fixture(name)
    nch = iso_streams.nch_of(kw)
    sp, sd = api.parse_like_cli(mp3, 4096, ISO_LSF)
    n = sp.shape[0]
    got = pairs_to_samples(emul_lsf(emul, sp, sd), n, nch)
    want = lsf_pcm_of_records(oracle.decode(sp, sd), n, nch)
    assert np.abs(got.astype(np.int32) - want).max() <= 1, "kernel source against the oracle"
    mx, _ = ffmpeg_error(got.astype(np.float64) * (32768.0 / 32767.0), theirs)
    assert mx <= iso_streams.LSF_TOL_S16_LSB[name], "%s: max %.2f LSB against FFmpeg" % (name, mx)
    # chunks of one and three record-frames (halos across pairs of LSF frames), an odd number of frames, float PCM
    for chunk in (1, 3):
        assert np.array_equal(pairs_to_samples(emul_lsf(emul, sp, sd, chunk), n, nch), got), chunk
    assert np.array_equal(pairs_to_samples(emul_lsf(emul, sp[:n - 1], sd[:n - 1]), n - 1, nch), got[:(n - 1) * 576])
    f = pairs_to_samples(emul_lsf(emul, sp[:9], sd[:9], f32=True), 9, nch)
    q = np.clip(np.trunc(f.astype(np.float64) * 32767.0), -32767, 32767)
    assert np.abs(q - got[:9 * 576]).max() <= 0


def test_a_bits_mode_scan_stops_at_the_first_lsf_frame():
    """the device Huffman stage reads MPEG-1 side info only: with PDMP3_ISO_LSF a bits-mode scan (host/frame_parse.c read_header:
    lsf_seen) ends at the first LSF Layer III frame -- the whole-stream decoder then hands the stream to its host-Huffman twin
    (host/bulk_api.c; GPU: tests/test_gpu_lsf.py) -- wherever that frame is; without the switch LSF frames are junk to every
    scan, as they are to the reference; and the count-only scan, which the decoder's caller sizes its buffer by, counts them"""
    from pdmp3_amd import api
    from pdmp3_amd.packer import packer
    lsf, _, _ = load_lsf_fixture("lsf_22k_ms")
    m1 = packer.generate(n_frames=20, seed=6, sfreq=1, mode=1, mode_ext=2, bitrate_index=9, iso_strict=True)
    only_m1, _, _ = api.parse_bits(m1, ISO_LSF)
    assert only_m1.shape[0] >= 17                                          # (an MPEG-1 stream is not touched by the switch; H10: its last frames stay in the ring)
    both, _, _ = api.parse_bits(m1 + lsf, ISO_LSF)
    assert only_m1.shape[0] <= both.shape[0] <= 20                          # the scan ended where the LSF frames begin
    tagged, _, _ = api.parse_bits(b"ID3" + bytes(40) + lsf, ISO_LSF)
    assert tagged.shape[0] == 0
    total, frames = api.scan_buffer(m1 + lsf, ISO_LSF)
    assert frames >= 20 + 28 and total > 20 * 4608 + 28 * 2304
    total0, frames0 = api.scan_buffer(m1 + lsf, 0)
    assert frames0 == both.shape[0]                                        # the reference's view: LSF headers are no headers
