#!/bin/bash
# whole-stream decoder to pageable memory: copy-out threads 4 / 6 / 8, interleaved
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --reps 6 "$@" 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5; do for t in 3 4 6 8; do echo -n "threads $t: "; run --threads $t; done; echo; done
