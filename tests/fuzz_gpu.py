#!/usr/bin/env python3
"""Randomised differential run on the GPU box (not collected by pytest: run by hand / tools/gpu_soak.sh):
random packer configurations -- rate, mode, mode extension, bit rate / VBR, CRC, block-type mix, mixed blocks, table 33,
gains, intensity cuts, MPEG-1 / LSF / 2.5 -- and a random switch mask; each stream through the product (whole-stream decoder
with device and with host Huffman, streaming API) against the oracle with the same mask: int16 PCM within 1 LSB of the
oracle's, the product's three paths bit-identical among themselves.  Prints every configuration that fails.

    python3 tests/fuzz_gpu.py [seconds] [seed] [corrupt]

`corrupt`: one to six random bit flips per stream.  What the reference does with those is mostly defined -- resyncs, CRC
bytes, the reservoir running dry (H9), stale state -- and compared like everything else; two classes are not and are only
required to get through the product: streams on which the reference replays its input ring for ever (DESIGN.md section 7;
pdmp3_amd_scan_buffer says so beforehand, and the whole-stream decoder must say the same) and streams that make its line
counter wrap (the oracle flags them).
"""
import faulthandler
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle                     # noqa: E402
from pdmp3_amd import api                            # noqa: E402
from pdmp3_amd.packer import packer                  # noqa: E402

ISO_LSF = 0x40


def random_cfg(rng):
    version = rng.choice([0, 0, 0, 1, 2])
    mode = rng.choice([0, 1, 1, 1, 2, 3])
    kw = dict(n_frames=rng.randint(20, 90), seed=rng.randint(1, 1 << 30), sfreq=rng.randint(0, 2), mode=mode,
              mode_ext=rng.randint(0, 3) if mode == 1 else 0, crc=rng.random() < 0.3,
              block_pct=rng.choice([(70, 10, 10, 10), (40, 10, 40, 10), (25, 25, 25, 25), (100, 0, 0, 0), (10, 10, 70, 10)]),
              mixed_pct=rng.choice([0, 30, 50, 100]), table33_pct=rng.choice([0, 0, 40, 100]), gain=rng.choice([(110, 150), (140, 165), (90, 120)]),
              version=version)
    if version:
        kw["bitrate_index"] = rng.randint(3, 14)
        kw["iso_strict"] = True                       # (LSF is decoded by the standard: what the reference-valid-only streams mean there is undefined)
        kw["is_cut_pct"] = rng.choice([0, 0, 40])
        kw["narrow_scales"] = True
        if version == 2 and kw["sfreq"] == 2:
            kw["mixed_pct"] = 0                       # (8 kHz: no mixed blocks)
            kw["bitrate_index"] = min(kw["bitrate_index"], 11)   # (128 kbps and up are frames of >= 1152 bytes there: the ring replay below)
    else:
        if rng.random() < 0.4:
            kw.update(vbr=True, vbr_lo=rng.randint(1, 6), vbr_hi=rng.randint(7, 14 if kw["sfreq"] != 2 else 12))
        else:
            kw["bitrate_index"] = rng.randint(4 if mode != 3 else 2, 14)
        if kw["sfreq"] == 2:                          # 32 kHz at 256 kbps = frames of 1152 bytes: the reference replays its ring for ever
            kw["bitrate_index"] = min(kw.get("bitrate_index", 12), 12)
            kw["vbr_hi"] = min(kw.get("vbr_hi", 12), 12)
        strict = rng.random() < 0.5
        kw["iso_strict"] = strict
        if strict:
            kw["is_cut_pct"] = rng.choice([0, 30, 60])
            kw["narrow_scales"] = rng.random() < 0.5
    return kw


def run(seconds, seed, corrupt, result):
    rng = random.Random(seed)
    orc = Oracle()
    log = open(os.environ.get("PDMP3_FUZZ_LOG", "/dev/null"), "w")
    t_end = time.time() + seconds
    n, bad, frames = 0, 0, 0
    kinds = {}
    while time.time() < t_end:
        kw = random_cfg(rng)
        iso = (rng.choice([0, 0x3f, rng.randint(0, 0x3f)]) | ISO_LSF) if kw["version"] else rng.choice([0, 0, 0x3f, rng.randint(0, 0x3f)])
        print("stream %d iso %#x %s" % (n, iso, json.dumps(kw)), file=log, flush=True)   # (the last line names the stream a hang is in)
        faulthandler.dump_traceback_later(120, exit=True)     # a stream that takes two minutes is a hang: say where, and stop
        try:
            mp3 = packer.generate(**kw)
        except AssertionError:
            faulthandler.cancel_dump_traceback_later()
            continue                                  # (a combination the packer does not make)
        if corrupt:
            b_ = bytearray(mp3)
            for _ in range(rng.randint(1, 6)):
                b_[rng.randrange(len(b_))] ^= 1 << rng.randrange(8)
            mp3 = bytes(b_)
            try:
                api.scan_buffer(mp3, iso)
            except api.RingReplay:                     # no finite reference output: the whole-stream decoder has to say so too
                said = 0
                for host_huffman in (False, True):
                    b = api.BulkDecoder(threads=2, window_frames=32, host_huffman=host_huffman)
                    try:
                        b.set_quirks(iso)
                        b.decode(mp3)
                    except api.RingReplay:
                        said += 1
                    finally:
                        b.close()
                faulthandler.cancel_dump_traceback_later()
                n += 1
                kinds["ring replay"] = kinds.get("ring replay", 0) + 1
                if said != 2:
                    bad += 1
                    print("FAIL the whole-stream decoder decoded a stream the scan calls a ring replay  iso %#x  %s" % (iso, json.dumps(kw)), flush=True)
                continue
        want = np.frombuffer(orc.decode_buffer_like_cli_iso(mp3, iso), dtype=np.int16)
        # a granule whose part2_3_length ends inside its scalefactors (the packer makes some at 32-48 kbps) wraps the reference's
        # line counter: its output is undefined from there on -- the product must get through it, nothing is compared
        undefined = orc.last_undefined
        outs = []
        why = None
        overloaded = False
        try:
            for host_huffman in (False, True):
                b = api.BulkDecoder(threads=2, window_frames=rng.choice([16, 32, 2048]), host_huffman=host_huffman)
                try:
                    b.set_quirks(iso)
                    outs.append(b.decode(mp3).reshape(-1))
                finally:
                    b.close()
            d = api.Decoder()
            try:
                d.set_quirks(iso)
                outs.append(np.frombuffer(api.decode_like_cli(mp3, d), dtype=np.int16))
            finally:
                d.close()
            # every few streams two more paths: the whole-stream decoder with the PCM left in device memory, and the streaming API's
            # float PCM, whose int16 is by definition clip(trunc(float x 32767)) (include/pdmp3.h) -- but for sums beyond 2^31 / 32767,
            # where the reference's conversion wraps (P:2028-2031)
            extra = None
            if n % 4 == 0 and outs[0].size:
                import torch
                b = api.BulkDecoder(threads=2, window_frames=rng.choice([32, 2048]))
                try:
                    b.set_quirks(iso)
                    t = torch.zeros(outs[0].size, dtype=torch.int16, device="cuda")
                    torch.cuda.synchronize()
                    b.decode_into_device(mp3, t)
                    dev = t.cpu().numpy()
                finally:
                    b.close()
                d = api.Decoder()
                try:
                    d.set_quirks(iso)
                    assert d.set_encoding(api.PDMP3_ENC_FLOAT_32) == 0
                    f32 = np.frombuffer(api.decode_like_cli(mp3, d), dtype=np.float32)
                finally:
                    d.close()
                if not np.array_equal(dev, outs[0]):
                    extra = "PCM left in device memory differs from PCM in host memory"
                elif f32.shape != outs[0].shape:
                    extra = "float PCM has %d samples, int16 PCM %d" % (f32.size, outs[0].size)
                else:
                    x = f32.astype(np.float64) * 32767.0
                    q = np.clip(np.trunc(x), -32767, 32767).astype(np.int32)
                    off = (q != outs[0]) & (np.abs(x) < 2147483000.0)
                    if off.any():
                        extra = "int16 PCM is not clip(trunc(float PCM x 32767)) at %d samples" % int(off.sum())
            if undefined:
                pass
            elif extra:
                why = extra
            elif not (outs[0].shape == outs[1].shape == outs[2].shape and np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])):
                why = "the product's paths differ: shapes %s" % [o.shape for o in outs]
            elif outs[0].shape != want.shape:
                why = "length %d against the oracle's %d" % (outs[0].size, want.size)
            else:
                diff = int(np.abs(outs[0].astype(np.int32) - want.astype(np.int32)).max()) if want.size else 0
                if diff > 1:
                    # a flipped bit in global_gain or a scalefactor makes signals thousands of times full scale, whose int16 is
                    # mostly the clip value -- and where it is not, the rounding of sums of such terms (the product fuses
                    # multiply-adds, the reference does not) is worth LSBs: the bar of 1 LSB is for signals an int16 can hold
                    # (tests/corpus.py gives its one loud corpus 32 LSB).  Loudness from the oracle's own float PCM of the records.
                    _, sp_t, sd_t = orc.decode_buffer_like_cli_iso(mp3, iso, tap_frames=kw["n_frames"] + 8)
                    _, f32 = orc.decode_f32(sp_t, sd_t)
                    peak = float(np.abs(f32).max()) if f32.size else 0.0
                    if peak > 4.0 and diff <= 64:
                        overloaded = True
                    else:
                        why = "max |diff| %d LSB against the oracle (float peak %.3g x full scale)" % (diff, peak)
        except Exception as e:                         # noqa: BLE001
            why = "exception %r" % (e,)
        faulthandler.cancel_dump_traceback_later()
        n += 1
        frames += kw["n_frames"]
        k = "undefined in the reference" if undefined else "overloaded (> 4 x full scale, within 64 LSB)" if overloaded else ("lsf%d" % kw["version"] if kw["version"] else "mpeg1") + ("/iso" if iso & 0x3f else "/ref")
        kinds[k] = kinds.get(k, 0) + 1
        if why:
            bad += 1
            print("FAIL %s  iso %#x  %s" % (why, iso, json.dumps(kw)), flush=True)
            dump = os.environ.get("PDMP3_FUZZ_DUMP")           # the stream's bytes as they were decoded (after the bit flips), to look at afterwards
            if dump and bad <= 40:
                open(os.path.join(dump, "fail_%d_iso%02x.mp3" % (bad, iso)), "wb").write(mp3)
    result.append((n, frames, kinds, bad))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261003
    corrupt = len(sys.argv) > 3 and sys.argv[3] == "corrupt"
    # $PDMP3_FUZZ_THREADS = N: N such loops at once in this process (seeds seed, seed + 1, ...): the decoders' pipelines, the
    # streaming API's shared helper threads and the engines' streams side by side (ctypes calls run without the interpreter's lock)
    nthreads = int(os.environ.get("PDMP3_FUZZ_THREADS", "1"))
    Oracle().decode_buffer_like_cli_iso(packer.generate(n_frames=4, seed=1), 0)       # (the oracle's tables, built once, before any thread reads them)
    results = []
    import threading
    ths = [threading.Thread(target=run, args=(seconds, seed + i, corrupt, results)) for i in range(nthreads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    n = sum(r[0] for r in results); frames = sum(r[1] for r in results); bad = sum(r[3] for r in results)
    kinds = {}
    for r in results:
        for k, v in r[2].items():
            kinds[k] = kinds.get(k, 0) + v
    print("fuzz_gpu: %d streams (%d frames) in %.0f s on %d thread(s), %s; failures: %d" % (n, frames, seconds, nthreads, json.dumps(kinds, sort_keys=True), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
