#!/usr/bin/env python3
"""Soak test of the whole-stream pipeline on the GPU (k_rows / k_unpack / k_merge / k_decode of six windows in flight):
a stream with every block type and scfsi, VBR, is decoded over and over with different window sizes and by several
decoders at once; every result must be the same bytes.

  python tools/soak_bulk.py [rounds]
"""
import hashlib
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdmp3_amd import api
from pdmp3_amd.packer import packer


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    mp3 = np.frombuffer(packer.generate(n_frames=20000, seed=91, vbr=True, block_pct=(40, 10, 40, 10), mixed_pct=30), dtype=np.uint8)
    ref = None
    ref_pcm = [None]
    bad = 0
    lock = threading.Lock()

    def work(window, reps):
        nonlocal ref, bad
        b = api.BulkDecoder(threads=2, window_frames=window)
        try:
            for _ in range(reps):
                pcm = b.decode(mp3)
                h = hashlib.sha256(pcm.tobytes()).hexdigest()
                with lock:
                    if ref is None:
                        ref = h
                        ref_pcm[0] = pcm.copy()
                    elif h != ref:
                        bad += 1
                        w = np.argwhere(pcm.reshape(-1) != ref_pcm[0].reshape(-1)).reshape(-1) if pcm.size == ref_pcm[0].size else np.array([-1])
                        print("soak_bulk: window %d: %d samples differ, frames %s ... %s, values %s / %s" % (
                            window, w.size, sorted(set((w[:2000] // 2304).tolist()))[:8], int(w[-1]) // 2304, pcm.reshape(-1)[w[:4]].tolist(), ref_pcm[0].reshape(-1)[w[:4]].tolist()), flush=True)
        finally:
            b.close()

    work(2048, 1)
    for r in range(rounds):
        ths = [threading.Thread(target=work, args=(w, 3)) for w in (2048, 777, 4096, 16)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    print("soak_bulk: %d rounds x 4 decoders x 3 decodes, mismatches: %d" % (rounds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
