#!/bin/bash
# the split scan for host destinations (default now) against the one-thread scan: tests, C3 pinned / pageable, C4 pageable, bench line
timeout 900 python3 -m pytest tests/test_gpu_bulk.py tests/test_gpu_corpus.py tests/test_gpu_api.py -x -q 2>&1 | tail -2
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 "$@" 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
c4() { timeout 300 python3 tools/bulk_bench.py --c4 $1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4; do
  echo -n "pinned split: "; run --pinned; echo -n " one-thread: "; PDMP3_BULK_SCAN_THREADS=0 run --pinned
  echo -n " | pageable split: "; run; echo -n " one-thread: "; PDMP3_BULK_SCAN_THREADS=0 run
  echo -n " | C4 pageable 1 dec split: "; c4 1; echo -n " one-thread: "; PDMP3_BULK_SCAN_THREADS=0 c4 1; echo -n " | 4 dec split: "; c4 4; echo -n " one-thread: "; PDMP3_BULK_SCAN_THREADS=0 c4 4; echo
done
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print('bench e2e', e['frames_per_s'], e['pcm_to_pinned_host']['frames_per_s'], e['pcm_left_in_hbm']['frames_per_s'], e['three_destinations_same_pcm'])"; done
