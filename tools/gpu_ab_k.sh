#!/bin/bash
# whole-stream decoder: the gather's non-temporal stores on / off, PCM left in HBM and to pinned memory, runs interleaved
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 "$@" 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5 6; do echo -n "device nt: "; run --device-out; echo -n " memcpy: "; PDMP3_BULK_GATHER_NT=0 run --device-out; echo -n " | pinned nt: "; run --pinned; echo -n " memcpy: "; PDMP3_BULK_GATHER_NT=0 run --pinned; echo; done
