#!/bin/bash
# GPU box: device Huffman kernels -- parity tests of the whole-stream path, then kernel durations per 2048-frame window.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_unpack.sh TAG'
TAG=${1:-unpack}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_bulk.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o u -- python3 tools/bulk_bench.py --frames ${FRAMES:-40000} --threads 4 --pinned --reps 2 > $OUT/prof.log 2>&1
cut -c1-60,150-260 $OUT/prof/u_kernel_stats.csv
tail -1 $OUT/prof.log | cut -c150-400
