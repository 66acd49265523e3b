#!/bin/bash
# whole-stream decoder, PCM left in HBM: the granule kernel in workgroups of 8 waves (84 KB of LDS: one fits a CU beside a
# workgroup of k_unpack, 79 KB) against 16 (160 KB: a CU to itself); runs interleaved; then the kernels' durations
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5; do echo -n "W=16: "; run; echo -n " W=8: "; PDMP3_HIP_GRAN_W=8 run; echo; done
export TMPDIR=/tmp
OUT=gpurun_out/w8
mkdir -p $OUT
for w in 16 8; do
PDMP3_HIP_GRAN_W=$w timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tl$w -o tl -- python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 4 --device-out > /dev/null 2> $OUT/tl$w.log; echo "W=$w"; python3 tools/bulk_timeline.py $OUT/tl$w/tl 2>&1 | cut -c1-250 | tail -6
done
