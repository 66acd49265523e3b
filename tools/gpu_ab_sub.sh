#!/bin/bash
# whole-stream decoder, PCM left in HBM: the short last window on / off, private windows of other sizes; runs interleaved
timeout 900 python3 -m pytest tests/test_gpu_bulk.py -x -q 2>&1 | tail -3
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5; do echo -n "default: "; run; echo -n " no tail split: "; PDMP3_BULK_TAIL_SPLIT=0 run; for s in 192 384; do echo -n " $s: "; PDMP3_BULK_SUB_FRAMES=$s run; done; echo; done
