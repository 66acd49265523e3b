#!/bin/bash
# The split scan's host side alone on the GPU box (no engine; pdmp3_amd_test_split_scan): how long a scanner takes per frame
# against the size of its private windows (two scanners: they are the bound then), and the whole scan with 8.
OUT=gpurun_out/scan_diag
mkdir -p $OUT
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from pdmp3_amd.packer import packer
mp3 = packer.generate(n_frames=137813, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14)
open("/tmp/c3.mp3", "wb").write(mp3)
PY
cat > /tmp/ss2.py <<'PY'
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from pdmp3_amd import api
mp3 = open("/tmp/c3.mp3", "rb").read()
a = np.frombuffer(mp3, dtype=np.uint8)
lib = api.load_library()
f = lib.pdmp3_amd_test_split_scan
f.restype = C.c_longlong
f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]
k = int(sys.argv[1]); w = int(sys.argv[2]); reps = int(sys.argv[3])
fr = C.c_longlong(0)
best = None
for i in range(reps):
    t0 = time.perf_counter(); r = f(a.ctypes.data_as(C.c_void_p), len(mp3), w, k, 0, None, 0, C.byref(fr)); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print("K=%d sub=%d: best %.2f ms = %.0f ns per frame and scanner" % (k, w, best * 1e3, best * 1e9 * k / fr.value))
PY
for k in 2 8; do for w in 128 256 512 1024 4096; do python3 /tmp/ss2.py $k $w 6; done; done
