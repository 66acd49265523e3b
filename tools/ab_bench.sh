#!/bin/bash
# A/B of engine builds on ONE box: every pdmp3_amd/variants/*.so through bench.py, twice, interleaved.
# Usage: gpurun --timeout 900 -- 'bash tools/ab_bench.sh TAG'
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
  for so in pdmp3_amd/variants/*.so; do
    n=$(basename $so .so)
    PDMP3_HIP_LIB=$PWD/$so timeout 300 python3 bench.py --no-cpu --no-e2e --steps 300 --warmup 30 > $OUT/$n.$rep.json 2>/dev/null
    python3 - <<PY
import json
try:
    d = json.load(open("$OUT/$n.$rep.json"))
    big = d.get("roofline_large_batch") or {}
    print("%-24s rep $rep  C2 %.2f us (%.1f M)  big %s ms" % ("$n", d["roofline"]["avg_launch_ms"] * 1e3, d["value"] / 1e6, big.get("avg_launch_ms")))
except Exception as e:
    print("$n", "failed", e)
PY
  done
done
