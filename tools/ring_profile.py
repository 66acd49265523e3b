"""Per-turn shader-clock stamps of the persistent granule kernel (k_decode_p).  usage: python tools/ring_profile.py [n_frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmp3_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
eng = pdmp3_amd.Engine(0)
lib = eng.lib
lib.pdmp3_hip_debug_profile_phases.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
prof = torch.zeros((2 * n, 12), dtype=torch.int64, device=eng.tdev)
for _ in range(3):
    rc = lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), -4, prof.data_ptr(), None)
    assert rc == 0, lib.pdmp3_hip_last_error()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), -4, prof.data_ptr(), None)
b.record()
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
names = ["turn start->decisions", "constants/commit/H5", "requant", "aa+imdct", "send tails", "wait+take tails", "overlap+matrix",
         "send rows", "window own", "wait+take rows", "window hist+pcm"]
cus = 256
per = max(8, (n + cus - 1) // cus)
print("n_frames %d  kernel %.2f us, %d frames per workgroup = %d turns" % (n, a.elapsed_time(b) * 1e3, per, per * 2 // 16))
d = p[:, 1:] - p[:, :-1]
ok = (p > 0).all(axis=1)
g = np.arange(2 * n)
place = (g - (g // (2 * per)) * (2 * per)) % 16
turn = (g - (g // (2 * per)) * (2 * per)) // 16
print("granules with all stamps: %d of %d" % (ok.sum(), p.shape[0]))
m0 = ok & (turn >= 2)
for k, nm in enumerate(names):
    x = d[m0, k]
    print("  %-24s median %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f ticks" % (nm, np.median(x), np.percentile(x, 10), np.percentile(x, 90), x.max()))
tot = p[:, 11] - p[:, 0]
print("  %-24s median %8.0f  p10 %8.0f  p90 %8.0f ticks" % ("turn", np.median(tot[m0]), np.percentile(tot[m0], 10), np.percentile(tot[m0], 90)))
# time between consecutive turns of the same wave (start to start) and the gap between the end of a turn and the next start
nxt = g + 16
valid = ok & (nxt < 2 * n) & ((nxt // (2 * per)) == (g // (2 * per)))
valid[valid] &= ok[nxt[valid]]
s2s = p[nxt[valid], 0] - p[g[valid], 0]
gap = p[nxt[valid], 0] - p[g[valid], 11]
print("  start to start of a wave's turns: median %.0f  p90 %.0f; gap end -> next start: median %.0f" % (np.median(s2s), np.percentile(s2s, 90), np.median(gap)))
print("  by place: median ticks of requant | aa+imdct | send tails | wait tails | send rows | wait rows | turn")
for pl in range(16):
    m = m0 & (place == pl)
    print("    place %2d: %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f %7.0f" % (pl, np.median(d[m, 2]), np.median(d[m, 3]), np.median(d[m, 4]), np.median(d[m, 5]),
                                                                      np.median(d[m, 7]), np.median(d[m, 9]), np.median(tot[m])))
# one workgroup's first turns, wave by wave (ticks since the workgroup's first stamp)
wg = 3
base = wg * 2 * per
t0 = p[base:base + 2 * per, 0].min()
print("  workgroup %d, start of turns 0..4 by place (k ticks since its first stamp):" % wg)
for pl in range(16):
    print("    place %2d: %s" % (pl, " ".join("%7.1f" % ((p[base + 16 * t + pl, 0] - t0) / 1e3) for t in range(min(5, per * 2 // 16)))))
