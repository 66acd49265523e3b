"""Launch time against launch size for the engine's two kernels (development, round 5): the granule kernel (the engine's
choice up to PDMP3_HIP_GRAN_MAX frames) and the chunk kernel (PDMP3_HIP_CHAIN=0 forces it), each in a child process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, pdmp3_amd
eng = pdmp3_amd.Engine(0)
for n in (1, 16, 128, 1024, 2048, 4096, 6144, 8192, 10240, 12288, 16384, 24576, 32768, 65536, 131072):
    sp, sd, pcm = eng.alloc_frames(n)
    eng.generate(0x5EED0000C5, 0, n, sp, sd)
    reps = 200 if n <= 16384 else 30
    for _ in range(reps // 4): eng.decode(sp, sd, pcm)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): eng.decode(sp, sd, pcm)
    b.record(); torch.cuda.synchronize()
    print("RESULT %%d %%.3f %%s" %% (n, a.elapsed_time(b) / reps * 1e3, eng.last_launch_kernel()[:14]))
""" % ROOT
res = {}
for name, env in (("engine's choice", {}), ("chunk kernel", {"PDMP3_HIP_CHAIN": "0"})):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith("RESULT"):
            _, n, us, k = l.split(None, 3)
            res.setdefault(int(n), {})[name] = (float(us), k)
print("%8s  %28s  %28s" % ("frames", "engine's choice: us (ns/frame)", "chunk kernel: us (ns/frame)"))
for n in sorted(res):
    a = res[n].get("engine's choice", (0, "")); b = res[n].get("chunk kernel", (0, ""))
    print("%8d  %10.2f (%6.2f) %-14s  %10.2f (%6.2f) %-14s" % (n, a[0], a[0] * 1e3 / n, a[1], b[0], b[0] * 1e3 / n, b[1]))
