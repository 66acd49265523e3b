#!/bin/bash
# whole-stream decoder, PCM left in HBM: hop parts that grow by half against the equal ones, runs interleaved; C4 with one decoder too
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
c4() { timeout 300 python3 tools/bulk_bench.py --c4 1 --device-out 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']/1e6,2), end=' ')"; }
for i in 1 2 3 4 5 6; do echo -n "growing: "; run; echo -n " equal: "; PDMP3_BULK_PREPASS_EQUAL=1 run; echo -n " | C4 growing: "; c4; echo -n " equal: "; PDMP3_BULK_PREPASS_EQUAL=1 c4; echo; done
PDMP3_BULK_TRACE=2 timeout 200 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 2 --device-out 2>&1 > /dev/null | grep -E "pre-pass in" | tail -1 | cut -c1-600
