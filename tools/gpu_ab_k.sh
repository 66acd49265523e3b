#!/bin/bash
# whole-stream decoder, PCM left in HBM: capped first windows on / off, runs interleaved
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5 6; do echo -n "ramp: "; run; echo -n " no ramp: "; PDMP3_BULK_RAMP=0 run; echo; done
PDMP3_BULK_TRACE=2 timeout 200 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 2 --device-out 2>&1 > /dev/null | grep -E "^  -> window|split scan" | tail -22 | cut -c1-200
