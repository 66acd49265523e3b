"""MPEG-2 LSF / MPEG-2.5 on the GPU (last third; PDMP3_ISO_LSF): every product path against FFmpeg's decode of
the same bytes (tests/golden/lsf_*.npz) -- the engine's LSF launch on records, the streaming API (int16 and float32 PCM),
the whole-stream decoder with the Huffman stage on the host and the default one (which hands an LSF stream to a
host-Huffman decoder of its own), the CLI.  Bars: tests/iso_streams.py (2 LSB float / 3 LSB int16; 24 kHz: 6 / 7, see there)."""
import os
import subprocess

import numpy as np
import pytest

import iso_streams
from test_lsf_pin import ISO_LSF, ffmpeg_error, load_lsf_fixture, lsf_pcm_of_records, pairs_to_samples

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(iso_streams.LSF_STREAMS))
def test_gpu_lsf_streaming_api_and_bulk_against_ffmpeg(name):
    from pdmp3_amd import api
    mp3, theirs, kw = load_lsf_fixture(name)
    nch = iso_streams.nch_of(kw)
    outs = {}
    for enc, dt in ((api.PDMP3_ENC_FLOAT_32, np.float32), (api.PDMP3_ENC_SIGNED_16, np.int16)):
        d = api.Decoder()
        try:
            d.set_quirks(ISO_LSF)
            assert d.set_encoding(enc) == 0
            outs[enc] = np.frombuffer(api.decode_like_cli(mp3, d), dtype=dt).reshape(-1, nch)
            rc, rate, ch, e = d.getformat()
            assert rc == 0 and rate == iso_streams.lsf_rate_of(kw) and ch == nch and e == enc
        finally:
            d.close()
    f32, s16 = outs[api.PDMP3_ENC_FLOAT_32], outs[api.PDMP3_ENC_SIGNED_16]
    mx, rms = ffmpeg_error(np.clip(f32.astype(np.float64) * 32768.0, -32768.0, 32767.0), theirs)
    assert mx <= iso_streams.LSF_TOL_F32_LSB[name] and rms <= iso_streams.RMS_LSB, "%s float PCM: max %.2f LSB, rms %.3f against FFmpeg" % (name, mx, rms)
    mx16, _ = ffmpeg_error(s16.astype(np.float64) * (32768.0 / 32767.0), theirs)
    assert mx16 <= iso_streams.LSF_TOL_S16_LSB[name], "%s int16 PCM: max %.2f LSB against FFmpeg" % (name, mx16)
    for host_huffman in (True, False):                    # False: the device-Huffman decoder passes LSF streams to a host-Huffman one
        b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
        try:
            b.set_quirks(ISO_LSF)
            got = b.decode(mp3).reshape(-1, nch)
            again = b.decode(mp3).reshape(-1, nch)
        finally:
            b.close()
        assert np.array_equal(got, s16), "whole-stream decoder (host_huffman=%s) != streaming API" % host_huffman
        assert np.array_equal(again, s16)


@pytest.mark.parametrize("name", ["lsf_22k_msis", "lsf_24k_ms", "lsf_16k_mono", "lsf_11k_msis", "lsf_12k_mono", "lsf_8k_msis"])
def test_gpu_engine_lsf_launch_on_records(engine, oracle, name):
    """pdmp3_hip_decode_lsf_frames against the oracle on the records: whole, odd counts, batches handed on through a state
    block (an odd batch leaves the state after ONE granule of a record-frame), float PCM; MPEG-1 launches in between"""
    import torch
    from pdmp3_amd import api
    mp3, theirs, kw = load_lsf_fixture(name)
    nch = iso_streams.nch_of(kw)
    sp, sd = api.parse_like_cli(mp3, 4096, ISO_LSF)
    n = sp.shape[0]
    want = lsf_pcm_of_records(oracle.decode(sp, sd), n, nch)
    dsp, dsd = engine.upload(sp, sd)
    pcm = torch.zeros(((n + 1) // 2, 2304), dtype=torch.int16, device=engine.tdev)
    engine.decode_lsf(dsp, dsd, pcm)
    assert "independent chunks" in engine.last_launch_kernel()          # (never the granule kernels: they hand on whole frames)
    torch.cuda.synchronize()
    got = pairs_to_samples(pcm.cpu().numpy(), n, nch)
    assert np.abs(got.astype(np.int32) - want).max() <= 1
    mx, _ = ffmpeg_error(got.astype(np.float64) * (32768.0 / 32767.0), theirs)
    assert mx <= iso_streams.LSF_TOL_S16_LSB[name]
    # batches of 7, 1, 10 and the rest through one state block
    st = engine.new_state()
    out, a = [], 0
    for k in (7, 1, 10, n - 18):
        p = torch.zeros(((k + 1) // 2, 2304), dtype=torch.int16, device=engine.tdev)
        engine.decode_lsf(dsp[a:a + k], dsd[a:a + k], p, state=st)
        torch.cuda.synchronize()
        out.append(pairs_to_samples(p.cpu().numpy(), k, nch))
        a += k
    assert np.array_equal(np.concatenate(out), got)
    pf = torch.zeros(((n + 1) // 2, 2304), dtype=torch.float32, device=engine.tdev)
    engine.decode_lsf(dsp, dsd, pf)
    torch.cuda.synchronize()
    q = np.clip(np.trunc(pairs_to_samples(pf.cpu().numpy(), n, nch).astype(np.float64) * 32767.0), -32767, 32767)
    assert np.abs(q - got).max() <= 0


def test_gpu_lsf_and_mpeg1_streams_through_one_handle(oracle):
    """an MPEG-1 stream, an LSF stream and an MPEG-1 stream again fed to ONE handle: the batches split at the version
    changes, the synthesis state runs on (SURVEY H12's per-handle state); == the oracle on the concatenation"""
    from pdmp3_amd import api
    from pdmp3_amd.packer import packer
    a = packer.generate(n_frames=12, seed=5, sfreq=0, mode=1, mode_ext=2, bitrate_index=9, iso_strict=True)
    b, _, _ = load_lsf_fixture("lsf_16k_ms")
    mp3 = a + b + a
    d = api.Decoder()
    try:
        d.set_quirks(0x7f)
        got = np.frombuffer(api.decode_like_cli(mp3, d), dtype=np.int16)
    finally:
        d.close()
    want = np.frombuffer(oracle.decode_buffer_like_cli_iso(mp3, 0x7f), dtype=np.int16)
    assert got.shape == want.shape and got.size > (24 * 1152 + 30 * 576) * 2 * 0.9
    assert np.abs(got.astype(np.int32) - want).max() <= 1


def test_gpu_whole_stream_decoder_meets_lsf_frames_behind_the_first_bytes(oracle):
    """the whole-stream decoder's device Huffman stage reads MPEG-1 side info only: an LSF stream goes to the decoder's host-Huffman
    twin -- also when it does not OPEN with an LSF header (found by tests/fuzz_gpu.py with a flipped bit in the first header: the
    dispatch looked at the stream's first four bytes only, the device-Huffman scan then refused every LSF header and decoded
    nothing while the count-only scan counted every frame).  A tag in front, and an MPEG-1 stream that goes on as LSF: device-
    and host-Huffman decoders and the streaming API give the same PCM, the oracle's within 1 LSB"""
    from pdmp3_amd import api
    from pdmp3_amd.packer import packer
    lsf, _, _ = load_lsf_fixture("lsf_22k_ms")
    m1 = packer.generate(n_frames=20, seed=6, sfreq=1, mode=1, mode_ext=2, bitrate_index=9, iso_strict=True)
    tag = b"ID3\x03\x00\x00\x00\x00\x00\x21" + bytes(33)                   # an (empty) ID3v2 tag: 43 bytes that are no frame
    for name, mp3 in (("tag + LSF", tag + lsf), ("MPEG-1 then LSF", m1 + lsf), ("LSF then MPEG-1 then LSF", lsf + m1 + lsf)):
        want = np.frombuffer(oracle.decode_buffer_like_cli_iso(mp3, 0x7f), dtype=np.int16)
        outs = []
        for host_huffman in (False, True):
            b = api.BulkDecoder(threads=2, window_frames=16, host_huffman=host_huffman)
            try:
                b.set_quirks(0x7f)
                outs.append(b.decode(mp3).reshape(-1))
                plain = b.decode(m1).reshape(-1)              # and an MPEG-1 stream right after it through the same decoder
            finally:
                b.close()
            assert np.abs(plain.astype(np.int32) - np.frombuffer(oracle.decode_buffer_like_cli_iso(m1, 0x7f), dtype=np.int16)).max() <= 1, name
        d = api.Decoder()
        try:
            d.set_quirks(0x7f)
            outs.append(np.frombuffer(api.decode_like_cli(mp3, d), dtype=np.int16))
        finally:
            d.close()
        assert outs[0].shape == want.shape and want.size > 30 * 576 * 2, name
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), name
        assert np.abs(outs[0].astype(np.int32) - want).max() <= 1, name


def test_cli_decodes_lsf_with_the_env_switch(tmp_path):
    mp3, theirs, kw = load_lsf_fixture("lsf_22k_stereo")
    path = tmp_path / "l.mp3"
    path.write_bytes(mp3)
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    for streaming in ("0", "1"):
        subprocess.check_call([cli, str(path)], timeout=120, env=dict(os.environ, PDMP3_CLI_ISO="0x40", PDMP3_CLI_STREAMING=streaming))
        got = np.frombuffer((tmp_path / "l.mp3.raw").read_bytes(), dtype=np.int16).reshape(-1, 2)
        (tmp_path / "l.mp3.raw").unlink()
        mx, _ = ffmpeg_error(got.astype(np.float64) * (32768.0 / 32767.0), theirs)
        assert mx <= iso_streams.TOL_S16_LSB, streaming
    # without the switch: the reference's answer to such a stream -- nothing
    subprocess.call([cli, str(path)], timeout=120, env={k: v for k, v in os.environ.items() if k != "PDMP3_CLI_ISO"})
    raw = tmp_path / "l.mp3.raw"
    assert not raw.exists() or raw.stat().st_size == 0
