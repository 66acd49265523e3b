#!/bin/bash
# GPU box: the whole-stream decoder's copy to pageable memory -- helper threads and pinned buffers kept on the GPU's NUMA node
# (default) against left to the scheduler (PDMP3_BULK_NUMA=0), non-temporal stores against memcpy (PDMP3_BULK_PLAIN_COPY=1),
# 2 .. 6 copy threads, three rounds interleaved; the pinned destination and the device destination beside it.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_copyout.sh TAG'
TAG=${1:-copyout}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
fmt='import json,sys
d=json.loads(sys.stdin.readline())
print("   ".join("%d thr %.2f M" % (r["threads"], r["frames_per_s"]/1e6) for r in d["runs"]))'
for rep in 1 2 3; do
  for numa in 1 0; do
    for plain in 0 1; do
      echo -n "rep $rep NUMA=$numa PLAIN_COPY=$plain pageable: "
      PDMP3_BULK_NUMA=$numa PDMP3_BULK_PLAIN_COPY=$plain timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 2,4,6 --reps 4 2>/dev/null | python3 -c "$fmt"
    done
    echo -n "rep $rep NUMA=$numa pinned: "; PDMP3_BULK_NUMA=$numa timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 2 --pinned --reps 4 2>/dev/null | python3 -c "$fmt"
    echo -n "rep $rep NUMA=$numa device: "; PDMP3_BULK_NUMA=$numa timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 2 --device-out --reps 4 2>/dev/null | python3 -c "$fmt"
  done
done 2>&1 | tee $OUT/copyout.txt
