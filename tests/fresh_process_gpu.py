#!/usr/bin/env python3
"""What only a process's first moments show (not collected by pytest; run on the GPU box): pdmp3_cli -- a fresh process every
time, both of its modes (whole-file decoder / the streaming loop) -- over a handful of streams of every kind, each many times,
every output against the oracle's (int16, 1 LSB).  The GPU suite lives in one long process: its first launches happen once;
round 6's allocator finding (profiles/r06_malloc_async_probe.txt) was a first-launches-only failure that this would have shown.

    python3 tests/fresh_process_gpu.py [runs per stream and mode = 25]
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle                     # noqa: E402
from pdmp3_amd.packer import packer                  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    streams = {
        "joint 44.1 320k, every block type": (packer.generate(n_frames=120, seed=41, sfreq=0, mode=1, mode_ext=2, bitrate_index=14, block_pct=(40, 10, 40, 10), mixed_pct=30), 0),
        "mono 32k VBR": (packer.generate(n_frames=150, seed=42, sfreq=2, mode=3, vbr=True, vbr_lo=2, vbr_hi=10), 0),
        "stereo 48k 128k + CRC, ISO switches": (packer.generate(n_frames=100, seed=43, sfreq=1, mode=0, mode_ext=0, bitrate_index=9, crc=True, iso_strict=True), 0x3f),
        "M/S + intensity 44.1 VBR, ISO switches": (packer.generate(n_frames=90, seed=44, sfreq=0, mode=1, mode_ext=3, vbr=True, vbr_lo=5, vbr_hi=13, iso_strict=True, is_cut_pct=40, narrow_scales=True), 0x3f),
        "LSF 22.05k stereo": (packer.generate(n_frames=70, seed=45, sfreq=0, mode=0, mode_ext=0, bitrate_index=10, version=1, iso_strict=True, narrow_scales=True), 0x40),
        "LSF 16k M/S + intensity": (packer.generate(n_frames=70, seed=46, sfreq=2, mode=1, mode_ext=3, bitrate_index=8, version=1, iso_strict=True, is_cut_pct=40, narrow_scales=True), 0x7f),
        "MPEG-2.5 11.025k mono": (packer.generate(n_frames=70, seed=47, sfreq=0, mode=3, mode_ext=0, bitrate_index=6, version=2, iso_strict=True, narrow_scales=True), 0x40),
        "the real encoder's clip": (open(os.path.join(ROOT, "tests", "golden", "clip_invalid_keypress.mp3"), "rb").read(), 0),
    }
    orc = Oracle()
    cli = os.path.join(ROOT, "pdmp3_amd", "pdmp3_cli")
    d = tempfile.mkdtemp()
    bad = total = 0
    for name, (mp3, iso) in streams.items():
        want = np.frombuffer(orc.decode_buffer_like_cli_iso(mp3, iso), dtype=np.int16)
        path = os.path.join(d, "s.mp3")
        open(path, "wb").write(mp3)
        for streaming in ("0", "1"):
            fails = 0
            for _ in range(runs):
                env = dict(os.environ, PDMP3_CLI_STREAMING=streaming)
                if iso:
                    env["PDMP3_CLI_ISO"] = hex(iso)
                subprocess.run([cli, path], env=env, timeout=120, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                raw = path + ".raw"
                got = np.fromfile(raw, dtype=np.int16) if os.path.exists(raw) else np.zeros(0, np.int16)
                if os.path.exists(raw):
                    os.unlink(raw)
                ok = got.shape == want.shape and (got.size == 0 or int(np.abs(got.astype(np.int32) - want).max()) <= 1)
                fails += not ok
                total += 1
            bad += fails
            print("%-42s %s: %d of %d fresh processes off" % (name, "streaming loop " if streaming == "1" else "whole-file    ", fails, runs), flush=True)
    print("fresh_process_gpu: %d runs, %d off" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
