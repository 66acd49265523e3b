#!/usr/bin/env python3
"""Soak test of the granule kernel (decode_core.h run_granule) on the GPU: many synthetic 2048-frame batches, each
decoded one granule per wave (tails and rows handed from wave to wave) and with independent 2-frame chunks (halo): PCM
and carried state must be bit-identical.  Also sizes that leave the last workgroup partly filled, and several host
threads launching on their own HIP streams at once.

  python tools/soak_chain.py [rounds]
"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdmp3_amd.hip import Engine


def one(eng, seed, n, stream=None):
    spectra, side, pcm = eng.alloc_frames(n)
    eng.generate(seed, 0, n, spectra, side)
    st1, st2 = eng.new_state(), eng.new_state()
    pcm2 = torch.empty_like(pcm)
    eng.decode(spectra, side, pcm, chunk_frames=1, state=st1)      # one granule per wave
    eng.decode(spectra, side, pcm2, chunk_frames=2, state=st2)     # independent chunks
    torch.cuda.synchronize()
    return bool(torch.equal(pcm, pcm2)) and bool(torch.equal(st1, st2))


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    eng = Engine()
    bad = 0
    for r in range(rounds):
        n = 2048 if r % 4 else [1, 2, 7, 8, 9, 63, 65, 1000, 2047, 4099, 12288][(r // 4) % 11]
        if not one(eng, 0x5EED000000 + r, n):
            bad += 1
            print("MISMATCH seed %d n %d" % (r, n))
    print("sequential: %d rounds, %d mismatches" % (rounds, bad))

    def worker(k, out):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        ok = True
        with torch.cuda.stream(s):
            for r in range(rounds // 8):
                ok = one(eng, 0xABC000 + 1000 * k + r, 2048) and ok
        out[k] = ok
    out = {}
    th = [threading.Thread(target=worker, args=(k, out)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    print("4 threads on their own streams:", out)
    sys.exit(1 if bad or not all(out.values()) else 0)


if __name__ == "__main__":
    main()
