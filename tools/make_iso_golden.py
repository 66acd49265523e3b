#!/usr/bin/env python3
"""Writes tests/golden/iso_*.npz and tests/golden/lsf_*.npz: what an independent ISO decoder (FFmpeg's mpegaudiodec, tools/ffmpeg_ref.py) decodes the
conforming packer streams of tests/iso_streams.py to.  BUILD CONTAINER ONLY (the decoder lives in the image's kaleido
wheel); the fixtures travel, this script's dependencies do not.

A fixture = {kwargs (JSON) of packer.generate, sha256 of the stream's bytes, rate, channels, pcm int16 [samples][channels]
(FFmpeg's own int16 samples: its fixed-point decoder rounds to int16 itself, full scale 32768), offset}.
`offset` = index of FFmpeg's first sample in OUR sample count: searched (tools/ffmpeg_ref.align), not assumed; 0 for every
packer stream (no Xing / LAME tag, Chromium's demuxer trims nothing then, and FFmpeg's first frame is the stream's first).

Also prints the oracle's distance to each fixture with all switches on and with each one off: the numbers quoted in
DESIGN.md section 4."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import iso_streams                                   # noqa: E402
from ffmpeg_ref import FFmpegRef, align              # noqa: E402
from oracle.oracle import Oracle                     # noqa: E402
from pdmp3_amd.packer import packer                  # noqa: E402


def oracle_f32(orc, mp3, iso, nch, lsf=False):
    """the oracle's binary32 PCM x 32768 (FFmpeg's full scale; the reference's own int16 uses 32767, P:2028), clipped:
    [channels][samples].  lsf: 576 sample-frames per frame (the first half of each frame's place)"""
    _, sp, sd = orc.decode_buffer_like_cli_iso(mp3, iso, tap_frames=4096)
    _, f32 = orc.decode_f32(sp, sd)
    n = sp.shape[0]
    spf = 576 if lsf else 1152
    x = f32[:, :spf * 2].reshape(n * spf, 2).T if nch == 2 else f32[:, :spf].reshape(1, -1)
    return np.clip(x * 32768.0, -32768.0, 32767.0)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    orc = Oracle()
    report = []
    with FFmpegRef() as ff:
        for name, kw in iso_streams.STREAMS.items():
            mp3 = packer.generate(**kw)
            nch, rate = iso_streams.nch_of(kw), iso_streams.rate_of(kw)
            t = ff.decode(mp3, rate, nch)[:nch]
            # Chromium hands out FFmpeg's int16 as float: s / 32768 for s < 0, s / 32767 for s > 0
            theirs = np.where(t < 0, t * 32768.0, t * 32767.0)
            k = np.round(theirs)
            assert np.abs(theirs - k).max() < 2e-3, "FFmpeg's output is not int16"
            theirs = k
            ours = oracle_f32(orc, mp3, 0x3f, nch)
            off, e = align(ours / 32768.0, theirs / 32768.0, max_shift=1200)
            assert off == 0, (name, off, e)
            m = min(ours.shape[1], theirs.shape[1])
            err = np.abs(ours[:, :m] - theirs[:, :m])
            row = {"stream": name, "frames_ffmpeg": theirs.shape[1] // 1152, "frames_ours": ours.shape[1] // 1152,
                   "peak_lsb": float(np.abs(theirs).max()), "all_on_max": float(err.max()), "all_on_rms": float(np.sqrt((err ** 2).mean()))}
            for bit, label in ((0x01, "TABLE33"), (0x02, "MS_BOUND"), (0x04, "IS_SHORT"), (0x08, "SF21"), (0x10, "SF12"), (0x20, "IS_BOUND")):
                o = oracle_f32(orc, mp3, 0x3f & ~bit, nch)
                row["without_" + label] = float(np.abs(o[:, :m] - theirs[:, :m]).max())
            report.append(row)
            np.savez_compressed(os.path.join(out_dir, name + ".npz"), kwargs=json.dumps(kw), sha256=hashlib.sha256(mp3).hexdigest(),
                                rate=rate, channels=nch, offset=off, pcm=k.T.astype(np.int16))
            print(json.dumps(row))
    # ---- a stream that is NOT the packer's: tests/golden/clip_invalid_keypress.mp3 (MathJax's accessibility click: a real encoder's
    # 44.1 kHz stereo 64 kbps stream behind an ID3 tag, with an Info frame).  Chromium trims the Info frame and the encoder delay:
    # `offset` = 1152 + 1105 = 2257 is found by the search, not assumed.
    with FFmpegRef() as ff:
        mp3 = open(os.path.join(out_dir, "clip_invalid_keypress.mp3"), "rb").read()
        t = ff.decode(mp3, 44100, 2)[:2]
        theirs = np.where(t < 0, t * 32768.0, t * 32767.0)
        k = np.round(theirs)
        assert np.abs(theirs - k).max() < 2e-3
        row = {"stream": "iso_clip_real", "samples_ffmpeg": int(k.shape[1])}
        for iso, label in ((0x3f, "all_on"), (0, "reference")):
            ours = oracle_f32(orc, mp3, iso, 2)
            off, e = align(ours / 32768.0, k / 32768.0, max_shift=3000)
            m = min(ours.shape[1] - off, k.shape[1])
            err = np.abs(ours[:, off:off + m] - k[:, :m])
            row.update({label + "_offset": int(off), label + "_max": float(err.max()), label + "_rms": float(np.sqrt((err ** 2).mean())), "samples_compared": int(m)})
        np.savez_compressed(os.path.join(out_dir, "iso_clip_real.npz"), sha256=hashlib.sha256(mp3).hexdigest(), rate=44100, channels=2,
                            offset=row["all_on_offset"], pcm=k.T.astype(np.int16))
        report.append(row)
        print(json.dumps(row))
    json.dump(report, open(os.path.join(ROOT, "profiles", "r06_iso_pin.json"), "w"), indent=1)
    # ---- MPEG-2 LSF / MPEG-2.5: the same for tests/iso_streams.py LSF_STREAMS (PDMP3_ISO_LSF; the switches are implied) ----
    report = []
    import ctypes as C
    with FFmpegRef() as ff:
        for name, kw in iso_streams.LSF_STREAMS.items():
            mp3 = packer.generate(**kw)
            nch, rate = iso_streams.nch_of(kw), iso_streams.lsf_rate_of(kw)
            t = ff.decode(mp3, rate, nch)[:nch]
            theirs = np.where(t < 0, t * 32768.0, t * 32767.0)
            k = np.round(theirs)
            assert np.abs(theirs - k).max() < 2e-3, "FFmpeg's output is not int16"
            ours = oracle_f32(orc, mp3, 0x40, nch, lsf=True)
            off, e = align(ours / 32768.0, k / 32768.0, max_shift=600)
            assert off == 0, (name, off, e)
            m = min(ours.shape[1], k.shape[1])
            err = np.abs(ours[:, :m] - k[:, :m])
            row = {"stream": name, "rate": rate, "frames_ffmpeg": k.shape[1] // 576, "frames_ours": ours.shape[1] // 576, "peak_lsb": float(np.abs(k).max()),
                   "max": float(err.max()), "rms": float(np.sqrt((err ** 2).mean()))}
            if rate == 24000:
                orc.lib.orc_debug_24k_330(1)
                o330 = oracle_f32(orc, mp3, 0x40, nch, lsf=True)
                orc.lib.orc_debug_24k_330(0)
                row["max_with_ffmpegs_330"] = float(np.abs(o330[:, :m] - k[:, :m]).max())
            report.append(row)
            np.savez_compressed(os.path.join(out_dir, name + ".npz"), kwargs=json.dumps(kw), sha256=hashlib.sha256(mp3).hexdigest(),
                                rate=rate, channels=nch, offset=off, pcm=k.T.astype(np.int16))
            print(json.dumps(row))
    json.dump(report, open(os.path.join(ROOT, "profiles", "r06_lsf_pin.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
