#!/usr/bin/env python3
"""Soak of the whole-stream decoder's split scan with the PCM left in device memory: streams of several lengths and kinds,
three decoders at once, every decode compared with the first (sha-256 of the PCM).   python tools/soak_device.py [seconds]"""
import hashlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pdmp3_amd import api
from pdmp3_amd.packer import packer


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    streams = [packer.generate(n_frames=n, seed=200 + i, vbr=bool(i & 1), block_pct=(40, 10, 40, 10), mixed_pct=30, bitrate_index=14 - i)
               for i, n in enumerate((3300, 9000, 20000, 60000))]
    streams = [np.frombuffer(s, dtype=np.uint8) for s in streams]
    sizes = [api.scan_buffer(s)[0] for s in streams]
    ref = {}
    bad = [0]
    count = [0]
    lock = threading.Lock()
    t_end = time.time() + seconds

    def work(j):
        b = api.BulkDecoder(threads=2)
        outs = [torch.empty(max(s, 2) // 2, dtype=torch.int16, device="cuda:0") for s in sizes]
        try:
            k = j
            while time.time() < t_end:
                i = k % len(streams)
                k += 1
                outs[i].zero_()
                torch.cuda.synchronize()                    # (the decoder's streams do not wait for torch's)
                got, _, _ = b.decode_into_device(streams[i], outs[i], wait=True)
                h = hashlib.sha256(outs[i].cpu().numpy().tobytes()).hexdigest()
                with lock:
                    count[0] += 1
                    if got != sizes[i] or ref.setdefault(i, h) != h:
                        bad[0] += 1
        finally:
            b.close()
    ts = [threading.Thread(target=work, args=(j,)) for j in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print("soak_device: %d decodes by 3 decoders in %.0f s, mismatches: %d" % (count[0], seconds, bad[0]))
    return 1 if bad[0] else 0


if __name__ == "__main__":
    sys.exit(main())
