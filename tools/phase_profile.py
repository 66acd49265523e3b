"""Per-phase shader-clock profile of the decode kernel (k_decode_prof).
usage: python tools/phase_profile.py [n_frames] [chunk_frames]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pdmp3_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = pdmp3_amd.Engine(0)
lib = eng.lib
lib.pdmp3_hip_debug_profile_phases.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
sp, sd, pcm = eng.alloc_frames(n)
eng.generate(0x5EED0000C5, 0, n, sp, sd)
if os.environ.get("PDMP3_PROFILE_MONO"):            # same records as mono frames: channel 1 idle
    sd[:, :, 7] = (sd[:, :, 7] & 0xF3) | 0x0C
    sd[:, 1, :7] = 0; sd[:, 3, :7] = 0; sd[:, 1, 8:] = 0; sd[:, 3, 8:] = 0
nchunks = (n + chunk - 1) // chunk
prof = torch.zeros((nchunks, 12), dtype=torch.int64, device=eng.tdev)
for _ in range(2):
    rc = lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), chunk, prof.data_ptr(), None)
    assert rc == 0, lib.pdmp3_hip_last_error()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
rc = lib.pdmp3_hip_debug_profile_phases(eng.h, sp.data_ptr(), sd.data_ptr(), n, pcm.data_ptr(), chunk, prof.data_ptr(), None)
b.record()
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.float64)
gran = p[:, 8].sum()
names = ["load", "facts", "requant", "fetch+aa", "imdct+dct32", "-", "commit+scales", "window+store"]   # (the phases between run_chunk's PD_TICK marks)
tot = p[:, :8].sum()
print("n_frames %d chunk %d chunks %d kernel %.3f ms  (s_memtime ticks)" % (n, chunk, nchunks, a.elapsed_time(b)))
for k, nm in enumerate(names):
    print("  %-9s %10.1f ticks/granule  %5.1f %%" % (nm, p[:, k].sum() / gran, 100 * p[:, k].sum() / tot))
print("  total     %10.1f ticks/granule-wave" % (tot / gran))
def st(x):
    return "min %.0f  median %.0f  p90 %.0f  max %.0f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
# (s_memtime counters of different XCDs are not synchronised: only per-wave differences mean anything)
print("  per wave, in ticks:")
print("    setup (tables -> LDS/registers, first prefetch) " + st(p[:, 11] - p[:, 10]))
print("    granule loop " + st(p[:, :8].sum(axis=1)))
print("    granules per wave " + st(p[:, 8]))
