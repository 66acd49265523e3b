#!/bin/bash
# GPU box (its host CPU): scan rate of stage A, and the same with the pool copy taken out (upper bound of what moving
# that copy off the scanning thread can give).  Usage: gpurun --timeout 600 -- 'bash tools/gpu_scan.sh TAG'
TAG=${1:-scan}
OUT=gpurun_out/$TAG
mkdir -p $OUT /tmp/v
python3 - <<PY
import sys
sys.path.insert(0, ".")
from pdmp3_amd.packer import packer
open("/tmp/c3.mp3", "wb").write(packer.generate(n_frames=40000, seed=0xC3, sfreq=0, mode=1, mode_ext=2, bitrate_index=14))
PY
gcc -O2 -Iinclude tools/ubench/scan_rate.c -o /tmp/scan_rate -Lpdmp3_amd -lpdmp3 -lpdmp3_hip -Wl,-rpath,$PWD/pdmp3_amd
lscpu | grep -i "model name\|^CPU(s)\|L2\|L3" | tee $OUT/scan.txt
echo "== as built" | tee -a $OUT/scan.txt
/tmp/scan_rate /tmp/c3.mp3 | tee -a $OUT/scan.txt
sed 's|  ring_take(id, b->res_dst + b->pool_tail, size);|  id->istart = (id->istart + size) % INBUF_SIZE; id->processed += size;|' pdmp3_amd/host/pdmp3_host.c > pdmp3_amd/host/_nocopy.c
gcc -O2 -fPIC -std=gnu11 -pthread -shared -o /tmp/v/libpdmp3.so pdmp3_amd/host/_nocopy.c -Lpdmp3_amd -lpdmp3_hip 2>&1 | grep -i " error"
echo "== without the pool copy" | tee -a $OUT/scan.txt
LD_LIBRARY_PATH=/tmp/v:$PWD/pdmp3_amd /tmp/scan_rate /tmp/c3.mp3 | tee -a $OUT/scan.txt
