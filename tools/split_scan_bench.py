"""Times the split scan's host side alone (the test hook pdmp3_amd_test_split_scan with no output buffer: pre-pass,
K scanners, stitch in order, nothing copied, no engine): how the scanners scale on this host.
  python3 tools/split_scan_bench.py [--frames 137813] [--reps 5]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=137813)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scanners", default="1,2,4,8,12,16")
    ap.add_argument("--windows", default="4096,8192")
    args = ap.parse_args()
    from pdmp3_amd.packer import packer
    from pdmp3_amd import api
    mp3 = packer.generate(n_frames=args.frames, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
    a = np.frombuffer(mp3, dtype=np.uint8)
    lib = api.load_library()
    f = lib.pdmp3_amd_test_split_scan
    f.restype = C.c_longlong
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]
    out = {"host_cpus": os.cpu_count(), "frames": args.frames, "runs": []}
    for w in [int(x) for x in args.windows.split(",")]:
        for k in [int(x) for x in args.scanners.split(",")]:
            best = None
            frames = C.c_longlong(0)
            for _ in range(args.reps):
                t0 = time.perf_counter()
                f(a.ctypes.data_as(C.c_void_p), len(mp3), w, k, 0, None, 0, C.byref(frames))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out["runs"].append({"window": w, "scanners": k, "ms": round(best * 1e3, 2), "frames": frames.value,
                                "M_frames_per_s": round(frames.value / best / 1e6, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
