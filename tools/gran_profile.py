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

# who are the slow waves?
sdh = sd.cpu().numpy()                       # [n, 4, 128] bytes
flags = sdh[:, :, 3]                         # per (frame, gr*2+ch)
short = ((flags & 0x04) != 0) & (((flags & 0x18) >> 3) == 2)
g_idx = np.arange(2 * n)
f_idx, gr_idx = g_idx >> 1, g_idx & 1
h5 = short[f_idx, 3] & (gr_idx == 1)
any_short = short[f_idx, 2 * gr_idx] | short[f_idx, 2 * gr_idx + 1]
W = 16 if 2 * n > 2048 else 8
place = g_idx % W
life = p[:, 11] - p[:, 0]
order = np.argsort(-life)
top = order[: max(8, len(order) // 20)]
print("slowest 5 %% of the waves: lifetime median %.0f; of them H5 %.0f %%, a short block in the granule %.0f %%, place 0 %.0f %%, place %d %.0f %%"
      % (np.median(life[top]), 100 * h5[top].mean(), 100 * any_short[top].mean(), 100 * (place[top] == 0).mean(), W - 1, 100 * (place[top] == W - 1).mean()))
print("all waves: H5 %.1f %%, short %.1f %%" % (100 * h5.mean(), 100 * any_short.mean()))
for nm, m in (("H5", h5), ("short, not H5", any_short & ~h5), ("long", ~any_short), ("place 0", place == 0), ("place last", place == W - 1)):
    if m.any():
        print("  %-14s n %5d  lifetime median %7.0f  p90 %7.0f  max %7.0f" % (nm, m.sum(), np.median(life[m]), np.percentile(life[m], 90), life[m].max()))
# phases of the long-block waves against the short-block ones
for nm, m in (("long", ~any_short), ("short", any_short & ~h5), ("H5", h5), ("place 0", (place == 0) & ~any_short), ("last", (place == W - 1) & ~any_short), ("middle", (place > 0) & (place < W - 1) & ~any_short)):
    print("  phases (median ticks) %-6s: %s" % (nm, " ".join("%.0f" % np.median(d[m, k]) for k in range(11))))

print("  by place in the workgroup (long-block granules): median ticks of requant | aa+imdct | wait tails | lifetime")
for pl in range(W):
    m = (place == pl) & ~any_short
    print("    place %2d: %6.0f %6.0f %6.0f %7.0f" % (pl, np.median(d[m, 2]), np.median(d[m, 3]), np.median(d[m, 5]), np.median(life[m])))
