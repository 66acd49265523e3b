#!/bin/bash
# The whole-stream pipeline after a change to its kernels or its host side: the GPU tests that cover it, the kernels'
# durations per window (rocprofv3), end to end with the PCM left in HBM.
OUT=gpurun_out/bulk_check
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_bulk.py tests/test_gpu_corpus.py tests/test_gpu_iso.py -x -q 2>&1 | tail -4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bulk_stats -o bulk -- python3 tools/bulk_bench.py --frames 40000 --threads 2 --reps 2 > $OUT/bulk_under_rocprof.json 2> $OUT/bulk_stats.log; echo "bulk stats rc=$?"
cut -c1-150 $OUT/bulk_stats/bulk_kernel_stats.csv
for i in 1 2 3; do timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-out', d['runs'][0]['frames_per_s'])"; done
PDMP3_BULK_TRACE=1 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2>&1 >/dev/null | grep "split scan" | tail -3 | cut -c1-330
for j in 1 2 4; do timeout 600 python3 tools/bulk_bench.py --c4 $j --device-out 2> /dev/null | tail -1 | cut -c100-260; done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['end_to_end'])[:700])"
PDMP3_BULK_TRACE=2 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out --reps 1 > /dev/null 2> $OUT/c4_trace.txt; grep -c . $OUT/c4_trace.txt
