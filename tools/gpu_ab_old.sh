#!/bin/bash
# pinned destination: this tree, this tree's host library on the session-start engine library, the session-start build; interleaved
run() { timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 "$@" 2> /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['runs'][0]['frames_per_s']/1e6,2), end=' ')"; }
OLD=$PWD/pdmp3_amd/variants_old
for i in 1 2 3 4 5; do
  echo -n "new: "; run --pinned; echo -n " new host + old engine: "; LD_LIBRARY_PATH=$OLD:$LD_LIBRARY_PATH run --pinned; echo -n " old: "; PDMP3_HOST_LIB=$OLD/libpdmp3.so run --pinned; echo
done
