#!/bin/bash
# GPU box: where the whole-stream decoder's time goes (device Huffman path): pageable / pinned destination, window sizes,
# worker threads, the scanning thread's waits (PDMP3_BULK_TRACE), compact pool input against full snapshot rows.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_e2e.sh TAG'
TAG=${1:-e2e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F=${FRAMES:-137813}
T=${THREADS:-2,4,8,16}
for W in 0 4096; do
  echo "== window $W pageable"; timeout 300 python3 tools/bulk_bench.py --frames $F --threads $T --window $W 2>&1 | tail -1
  echo "== window $W pinned";   timeout 300 python3 tools/bulk_bench.py --frames $F --threads 2 --window $W --pinned 2>&1 | tail -1
done 2>&1 | tee $OUT/e2e.txt
echo "== trace, pageable, 4 threads" | tee -a $OUT/e2e.txt
PDMP3_BULK_TRACE=1 timeout 300 python3 tools/bulk_bench.py --frames $F --threads 4 2>&1 | tail -4 | tee -a $OUT/e2e.txt
echo "== snapshot rows, pinned" | tee -a $OUT/e2e.txt
PDMP3_BULK_SNAPSHOT_ROWS=1 timeout 300 python3 tools/bulk_bench.py --frames $F --threads 2 --pinned 2>&1 | tail -1 | tee -a $OUT/e2e.txt
echo "== rocprof kernel stats (pinned)" | tee -a $OUT/e2e.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o e2e -- python3 tools/bulk_bench.py --frames $F --threads 2 --pinned --reps 1 > $OUT/prof.log 2>&1
cut -c1-200 $OUT/prof/e2e_kernel_stats.csv | tee -a $OUT/e2e.txt
