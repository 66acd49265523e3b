#!/usr/bin/env python3
"""The persistent granule kernel (k_decode_p, chunk_frames = -3) against the engine's own choice (k_decode_g up to 12288
frames, k_decode beyond) on one GPU: PCM and carried state bit-identical, launch time per size.

  python tools/ring_bench.py [sizes...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdmp3_amd.hip import Engine

RING = -3


def timed(eng, sp, sd, pcm, chunk, reps):
    for _ in range(3):
        eng.decode(sp, sd, pcm, chunk_frames=chunk)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        eng.decode(sp, sd, pcm, chunk_frames=chunk)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [16, 100, 2048, 2049, 4096, 8192, 12288, 16384, 32768, 65536, 125002, 131072]
    eng = Engine()
    bad = 0
    for n in sizes:
        sp, sd, pcm = eng.alloc_frames(n)
        eng.generate(0x5EED0000C5, 1000, n, sp, sd)
        pcm2 = torch.empty_like(pcm)
        s1, s2 = eng.new_state(), eng.new_state()
        eng.decode(sp, sd, pcm, chunk_frames=0, state=s1)
        kind0 = eng.last_launch_kernel()
        eng.decode(sp, sd, pcm2, chunk_frames=RING, state=s2)
        torch.cuda.synchronize()
        ok = bool(torch.equal(pcm, pcm2)) and bool(torch.equal(s1, s2))
        bad += not ok
        reps = 200 if n <= 4096 else 30
        t0 = timed(eng, sp, sd, pcm, 0, reps)
        t1 = timed(eng, sp, sd, pcm2, RING, reps)
        print("%7d frames: %-8s %9.1f us (%s) | persistent %9.1f us = %6.1f M frames/s, frac %.4f" % (
            n, "equal" if ok else "MISMATCH", t0, kind0.split(" ")[0], t1, n / t1, n * 9728 / (t1 * 1e-6) / 8e12))
        del sp, sd, pcm, pcm2
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
