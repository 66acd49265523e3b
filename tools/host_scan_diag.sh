#!/bin/bash
# The split scan's host side alone on the GPU box (no engine): per private window when it was taken, when its snapshot was
# there, when it was scanned and when the stitcher got it -- who waits for whom.  Then the whole-stream decoder with the
# PCM left in HBM (what the scan feeds).
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
for i in range(reps):
    t0 = time.perf_counter(); r = f(a.ctypes.data_as(C.c_void_p), len(mp3), w, k, 0, None, 0, C.byref(fr)); dt = time.perf_counter() - t0
    print("RUN", k, w, r, fr.value, round(dt * 1e3, 3), "ms", file=sys.stderr)
PY
for k in ${KS:-8 12 16}; do
  PDMP3_BULK_TRACE=2 python3 /tmp/ss2.py $k 1024 6 > /dev/null 2> $OUT/trace_k$k.txt
  echo "K=$k:" $(grep RUN $OUT/trace_k$k.txt | awk '{print $6}')
  grep "pre-pass in" $OUT/trace_k$k.txt | sed 's/;.*//' | tail -3
done
echo "no spin:"; PDMP3_BULK_SCAN_SPIN=0 python3 /tmp/ss2.py 8 1024 6 2>&1 > /dev/null | grep RUN | awk '{print $6}' | tr '\n' ' '; echo
for w in 512 2048; do echo "sub $w:" $(python3 /tmp/ss2.py 8 $w 6 2>&1 > /dev/null | grep RUN | awk '{print $6}'); done
for t in 8 12 16; do
  echo "decoder, scanners $t:"
  PDMP3_BULK_TRACE=1 PDMP3_BULK_SCAN_THREADS=$t timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> $OUT/bulk_t$t.err | tail -1 | cut -c1-400
  grep "split scan" $OUT/bulk_t$t.err | tail -2 | cut -c1-330
done
echo "decoder, default:"; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | cut -c1-400
lscpu | head -25 > $OUT/lscpu.txt
