"""Per-wave shader-clock stamps of the granule kernel (k_decode_g).  usage: python tools/gran_profile.py [n_frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmp3_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
eng = pdmp3_amd.Engine(0)
lib = eng.lib
lib.pdmp3_hip_debug_profile_phases.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C2, 0, n, sp, sd)
prof = torch.zeros((2 * n, 12), dtype=torch.int64, device=eng.tdev)
for _ in range(3):
    rc = lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), -2, prof.data_ptr(), None)
    assert rc == 0, lib.pdmp3_hip_last_error()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), -2, prof.data_ptr(), None)
b.record()
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
names = ["entry->tables+ticket", "prefetch/commit/scales", "requant", "aa+imdct", "publish tails", "wait+take tails", "overlap+matrix",
         "publish rows", "window own", "wait+take rows", "window hist+pcm"]
print("n_frames %d  kernel %.2f us" % (n, a.elapsed_time(b) * 1e3))
d = p[:, 1:] - p[:, :-1]
ok = (p > 0).all(axis=1)
print("waves with all stamps: %d of %d" % (ok.sum(), p.shape[0]))
for k, nm in enumerate(names):
    x = d[ok, k]
    print("  %-24s median %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f ticks" % (nm, np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max()))
tot = p[ok, 11] - p[ok, 0]
print("  %-24s median %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f ticks" % ("wave lifetime", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90), tot.max()))
# per-XCD clocks are not synchronised; but within the launch the spread of entry times on one counter is informative
print("  entry spread (max - min over all waves, unsynchronised counters): %.0f ticks" % (p[ok, 0].max() - p[ok, 0].min()))
