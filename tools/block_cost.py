#!/usr/bin/env python3
"""What block switching costs a C2 launch: the generated batch with its block flags edited on the host (same spectra),
five interleaved repetitions of 300 launches per variant.   python tools/block_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pdmp3_amd

eng = pdmp3_amd.Engine(0)
n = 2048
LONG = np.uint8(0xff ^ (0x04 | 0x18 | 0x20))


def make(fn):
    sp, sd, pcm = eng.alloc_frames(n)
    eng.generate(0x5EED0000C2, 0, n, sp, sd)
    h = sd.cpu().numpy().copy().view(np.uint8).reshape(n, 4, 128)
    fn(h)
    sd.view(torch.uint8).reshape(n, 4, 128).copy_(torch.from_numpy(h).to(sd.device))
    return sp, sd, pcm


def timeit(b):
    sp, sd, pcm = b
    for _ in range(10):
        eng.decode(sp, sd, pcm)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        eng.decode(sp, sd, pcm)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / 300


def is_short(f):
    return ((f & 4) != 0) & (((f >> 3) & 3) == 2)


def keep_h5(k):
    def fn(h):
        idx = np.nonzero(is_short(h[:, 3, 3]))[0]
        if k >= 0:
            h[idx[k:], 3, 3] &= LONG
    return fn


def no_mixed(h): h[:, :, 3] &= np.uint8(0xff ^ 0x20)
def no_short(h):
    f = h[:, :, 3]; f[is_short(f)] &= LONG
def no_ws(h): h[:, :, 3] &= LONG
def all_short(h):
    h[:, :, 3] &= np.uint8(0xff ^ (0x18 | 0x20)); h[:, :, 3] |= np.uint8(0x04 | (2 << 3))


variants = [("as generated", keep_h5(-1)), ("mixed flag cleared", no_mixed), ("no H5 frames (granule 1 / channel 1 short -> long)", keep_h5(0)),
            ("32 H5 frames kept", keep_h5(32)), ("no short blocks", no_short), ("no window switching", no_ws), ("every block short", all_short)]
bufs = [(nm, make(fn)) for nm, fn in variants]
res = {nm: [] for nm, _ in variants}
for rep in range(5):
    for nm, b in bufs:
        res[nm].append(timeit(b))
for nm, _ in variants:
    print("%-52s %s  median %.2f us" % (nm, " ".join("%.2f" % x for x in res[nm]), float(np.median(res[nm]))))
