"""How does the chunk kernel's time per wave change with the number of waves a SIMD holds?  (development, round 5)

Runs the 131072-frame launch with PDMP3_HIP_DEBUG_LDS_PAD = 0 (two waves per SIMD: 15.5 KB of LDS and 256 registers per
wave) and with pads that leave a CU 4 and 6 waves, each in a child process (the pad is read once).  If the time per
launch doubles with half the waves, the waves do not get in each other's way (a wave's own latency chain is the bound
and a third wave per SIMD would pay); if it stays, the SIMD is the bound."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch, pdmp3_amd
eng = pdmp3_amd.Engine(0)
n = 131072
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
for _ in range(5): eng.decode(sp, sd, pcm, chunk_frames=int(os.environ.get("CHUNK", "32")))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): eng.decode(sp, sd, pcm, chunk_frames=int(os.environ.get("CHUNK", "32")))
b.record(); torch.cuda.synchronize()
print("RESULT %%.4f" %% (a.elapsed_time(b) / 20))
""" % ROOT

for pad, what in ((0, "8 waves per CU (2 per SIMD)"), (8192, "6 waves per CU"), (24576, "4 waves per CU (1 per SIMD)"), (65536, "2 waves per CU")):
    for chunk in (32, 16):
        env = dict(os.environ, PDMP3_HIP_DEBUG_LDS_PAD=str(pad), CHUNK=str(chunk))
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        ms = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("pad %6d  %-32s chunk %2d frames: %s ms per 131072-frame launch" % (pad, what, chunk, ms[0] if ms else "failed: " + r.stderr[-300:]))
