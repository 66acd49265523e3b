#!/usr/bin/env python3
"""The drop-in streaming API (pdmp3_feed / pdmp3_read, include/pdmp3.h) on a C3-style stream, driven by the C loop
pdmp3_amd_stream_loop: frames/s at the reference driver's cadence and with the ring kept full.
  python3 tools/stream_api_bench.py [frames]          (PDMP3_STREAM_THREADS = helper threads of the read-ahead batches)"""
import json, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdmp3_amd import api
from pdmp3_amd.packer import packer

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mp3 = np.frombuffer(packer.generate(n_frames=nf, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14), dtype=np.uint8)
out = {"helpers": os.environ.get("PDMP3_STREAM_THREADS", "default"), "spin": os.environ.get("PDMP3_STREAM_SPIN", "default")}
for key, (feed, read, eager) in (("reference_cadence", (4096, 16384, False)), ("ring_kept_full", (4096, 65536, True))):
    api.stream_loop(mp3[:200000], feed, read, eager, want_pcm=False)
    best = 1e9
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(3):
        t0 = time.perf_counter()
        nbytes, _ = api.stream_loop(mp3, feed, read, eager, want_pcm=False)
        best = min(best, time.perf_counter() - t0)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    out[key] = round(nbytes / 4608.0 / best, 1)
    # CPU time of the whole process (the caller + the helper threads, user + system) per decoded frame, microseconds
    out[key + "_cpu_us_per_frame"] = round(((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) / (3 * nbytes / 4608.0) * 1e6, 2)
print(json.dumps(out))
