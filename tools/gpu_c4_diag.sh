#!/bin/bash
# C4 with one decoder: where the time goes; the link's H2D rate
OUT=gpurun_out/c4_diag
mkdir -p $OUT
for i in 1 2; do BULK_BENCH_VERBOSE=1 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>&1 | grep -E "decoder 0|frames_per_s" | cut -c1-330; done
echo "one-thread scan:"; PDMP3_BULK_SCAN_THREADS=0 BULK_BENCH_VERBOSE=1 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>&1 | grep -E "decoder 0|frames_per_s" | cut -c100-330
echo "4 scanners:"; PDMP3_BULK_SCAN_THREADS=4 BULK_BENCH_VERBOSE=1 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>&1 | grep -E "decoder 0|frames_per_s" | cut -c100-330
echo "16 scanners:"; PDMP3_BULK_SCAN_THREADS=16 BULK_BENCH_VERBOSE=1 timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>&1 | grep -E "decoder 0|frames_per_s" | cut -c100-330
timeout 300 python3 tools/h2d_bw.py | tee $OUT/h2d_bw.json
