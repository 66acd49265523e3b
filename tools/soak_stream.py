#!/usr/bin/env python3
"""Soak of the streaming API with several handles at once on the GPU (include/pdmp3.h: each handle has its own engine
stream; the helper threads that decode the batches' main data are shared, a slot per batch): H threads run the C feed /
read loop (pdmp3_amd_stream_loop) over streams of different kinds and cadences for a while; every decode must give the
PCM the same stream gave when it was decoded alone (sha-256).   python tools/soak_stream.py [seconds] [handles]"""
import hashlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdmp3_amd import api
from pdmp3_amd.packer import packer


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    handles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    streams = [np.frombuffer(packer.generate(n_frames=n, seed=300 + i, vbr=bool(i & 1), block_pct=(40, 10, 40, 10), mixed_pct=30,
                                             mode=(1, 3, 0)[i % 3], mode_ext=2 if i % 3 == 0 else 0, bitrate_index=14 - 2 * i), dtype=np.uint8)
               for i, n in enumerate((1500, 4000, 2500, 6000))]
    cadences = [(4096, 16384, False), (4096, 65536, True), (2048, 40000, True), (4096, 4608, False)]
    want = {}
    for i, s in enumerate(streams):
        for j, (fb, rb, eager) in enumerate(cadences):
            n, pcm = api.stream_loop(s, fb, rb, eager)
            want[i, j] = (n, hashlib.sha256(pcm.tobytes()).hexdigest())
    bad, done = [0], [0]
    lock = threading.Lock()
    t_end = time.time() + seconds

    def run(h):
        k = h
        while time.time() < t_end:
            i, j = k % len(streams), (k // len(streams) + h) % len(cadences)
            fb, rb, eager = cadences[j]
            n, pcm = api.stream_loop(streams[i], fb, rb, eager)
            ok = (n, hashlib.sha256(pcm.tobytes()).hexdigest()) == want[i, j]
            with lock:
                done[0] += 1
                if not ok:
                    bad[0] += 1
            k += 1

    ths = [threading.Thread(target=run, args=(h,)) for h in range(handles)]
    for t in ths: t.start()
    for t in ths: t.join()
    print("soak_stream: %d decodes by %d handles at once in %.0f s, mismatches: %d" % (done[0], handles, seconds, bad[0]))
    return 1 if bad[0] else 0


if __name__ == "__main__":
    sys.exit(main())
