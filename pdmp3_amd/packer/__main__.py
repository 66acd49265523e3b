"""Corpus generator for the bitstream-level workloads of SURVEY.md 8d (there is no network for real files):

  python -m pdmp3_amd.packer c1 out/            one 44.1 kHz stereo 128 kbps CBR file
  python -m pdmp3_amd.packer c3 out/            1 h of 44.1 kHz joint stereo 320 kbps CBR (137 813 frames, ~144 MB)
  python -m pdmp3_amd.packer c4 out/            the mixed corpus: {mono, stereo, joint-MS} x {32, 44.1, 48 kHz} x
                                            {CBR, VBR} x {long, start/short/stop, mixed}, 64 files x >= 4096 frames
  python -m pdmp3_amd.packer custom out/ --frames 500 --sfreq 1 --mode 3 --vbr ...

Every file is a valid MPEG-1 Layer III stream (frame sync, side info, bit reservoir with main_data_begin, scale-
factors, Huffman-coded spectra from all code books incl. linbits and both count1 tables); the spectra are synthetic
(C2-style generator), so the audio is noise -- the point is the bitstream.  A manifest.json with sizes and SHA-256
is written beside the files.  32 kHz streams stop at 224 kbps: 256 kbps gives 1152-byte frames, which drive the
reference decoder into replaying its input ring (include/pdmp3_bulk.h, PDMP3_BULK_REPLAY).
"""
import argparse
import hashlib
import json
import os
import sys

from . import packer


def c4_specs(frames=4096):
    specs, seed = [], 400
    for mode, mext in ((3, 0), (0, 0), (1, 2)):
        for sfreq in (0, 1, 2):
            for vbr in (False, True):
                for blocks in ((100, 0, 0, 0), (40, 20, 20, 20), (20, 10, 60, 10)):
                    seed += 1
                    hi = 12 if sfreq == 2 else 14
                    specs.append(dict(n_frames=frames + seed % 64, seed=seed, sfreq=sfreq, mode=mode, mode_ext=mext, vbr=vbr,
                                      vbr_lo=4, vbr_hi=hi, bitrate_index=min(12, hi), block_pct=blocks,
                                      mixed_pct=50 if blocks[2] else 0))
    return specs + specs[:10]


def main():
    ap = argparse.ArgumentParser(prog="python -m pdmp3_amd.packer", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("what", choices=["c1", "c3", "c4", "custom"])
    ap.add_argument("outdir")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sfreq", type=int, default=0, help="0 = 44.1 kHz, 1 = 48 kHz, 2 = 32 kHz")
    ap.add_argument("--mode", type=int, default=1, help="0 stereo, 1 joint, 2 dual, 3 mono")
    ap.add_argument("--mode-ext", type=int, default=2, help="bit 1 = MS stereo")
    ap.add_argument("--bitrate-index", type=int, default=14)
    ap.add_argument("--vbr", action="store_true")
    ap.add_argument("--crc", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.outdir, exist_ok=True)
    if args.what == "c1":
        jobs = [("c1_128k.mp3", dict(n_frames=args.frames or 2000, seed=0xC1, sfreq=0, mode=1, mode_ext=2, bitrate_index=9))]
    elif args.what == "c3":
        jobs = [("c3_320k_1h.mp3", dict(n_frames=args.frames or 137813, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14))]
    elif args.what == "c4":
        jobs = [("c4_%02d.mp3" % i, s) for i, s in enumerate(c4_specs(args.frames or 4096))]
    else:
        jobs = [("custom.mp3", dict(n_frames=args.frames or 100, seed=args.seed, sfreq=args.sfreq, mode=args.mode,
                                    mode_ext=args.mode_ext, bitrate_index=args.bitrate_index, vbr=args.vbr, crc=args.crc))]
    manifest = []
    for name, spec in jobs:
        data = packer.generate(**spec)
        with open(os.path.join(args.outdir, name), "wb") as f:
            f.write(data)
        manifest.append({"file": name, "bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                         "spec": {k: (list(v) if isinstance(v, tuple) else v) for k, v in spec.items()}})
        print("%s  %d bytes" % (name, len(data)), file=sys.stderr)
    with open(os.path.join(args.outdir, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
