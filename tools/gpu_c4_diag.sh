#!/bin/bash
# C4 with one decoder: where the time goes; then the long stream with the PCM left in HBM
OUT=gpurun_out/c4_diag
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_corpus.py -x -q 2>&1 | tail -1
for i in 1 2 3; do BULK_BENCH_VERBOSE=1 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>&1 | grep -E "decoder 0|frames_per_s" | cut -c1-118,185-250 | tr '\n' ' '; echo; done
for j in 2 4; do timeout 300 python3 tools/bulk_bench.py --c4 $j --device-out 2>/dev/null | tail -1 | cut -c100-200; done
for i in 1 2 3; do timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-out', d['runs'][0]['frames_per_s'])"; done
PDMP3_BULK_TRACE=2 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out --reps 1 > /dev/null 2> $OUT/c4_trace.txt; grep "split scan" $OUT/c4_trace.txt | sed -n 20,24p | cut -c1-330
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/bulk_tl -o tl -- python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 4 --device-out > /dev/null 2> $OUT/bulk_tl.log; python3 tools/bulk_timeline.py $OUT/bulk_tl/tl > $OUT/bulk_timeline.txt 2>&1; cat $OUT/bulk_timeline.txt | cut -c1-260
