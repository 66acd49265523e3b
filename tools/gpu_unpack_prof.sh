#!/bin/bash
# development: where k_unpack's time goes (PDMP3_HIP_UNPACK_PROF=1 stamps in the kernel, one line per launch on stderr)
TAG=${1:-uprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PDMP3_HIP_UNPACK_PROF=1 timeout 300 python3 tools/bulk_bench.py --frames 20000 --threads 2 --reps 1 > $OUT/run.json 2> $OUT/prof.txt; echo "rc=$?"
grep "k_unpack prof" $OUT/prof.txt | tail -6
