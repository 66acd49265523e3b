#!/bin/bash
# Device-Huffman path on the GPU box: its parity tests, then kernel durations per 2048-frame window (rocprofv3) and the
# whole-stream rate.  Usage (from the repo root): gpurun --timeout 900 -- 'bash tools/gpu_unpack.sh TAG'
TAG=${1:-unpack}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_bulk.py tests/test_gpu_corpus.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bulk_stats -o bulk -- python3 tools/bulk_bench.py --frames 40000 --threads 2 --reps 2 > $OUT/bulk_under_rocprof.json 2> $OUT/bulk_stats.log; echo "bulk stats rc=$?"
cut -c1-150 $OUT/bulk_stats/bulk_kernel_stats.csv | head -8
timeout 600 python3 tools/bulk_bench.py --frames 137813 --threads 2,4 2> /dev/null | tail -1 > $OUT/bulk_decode.json; cat $OUT/bulk_decode.json
PDMP3_HIP_UNPACK_PROF=1 timeout 300 python3 tools/bulk_bench.py --frames 20000 --threads 2 --reps 1 2>&1 > /dev/null | grep "k_unpack prof" | tail -2
