#!/bin/bash
# whole-stream decoder towards host memory (split scan, 4 scanners): the engine's window size, interleaved
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 "$@" 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4; do
  for w in 2048 3072 4096 6144; do echo -n "pinned w$w: "; run --pinned --window $w; done; echo -n " | "
  for w in 2048 3072 4096 6144; do echo -n "pageable w$w: "; run --window $w; done; echo
done
