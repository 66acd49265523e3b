#!/bin/bash
# A/B of engine builds on ONE box: every pdmp3_amd/variants/*.so through bench.py, REPS times, interleaved.
# Usage: gpurun --timeout 900 -- 'bash tools/ab_bench.sh TAG [REPS]'
TAG=${1:-ab}
REPS=${2:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for so in pdmp3_amd/variants/*.so; do
    n=$(basename $so .so)
    PDMP3_HIP_LIB=$PWD/$so timeout 300 python3 bench.py --no-cpu --no-e2e --steps 300 --warmup 30 > $OUT/$n.$rep.json 2>$OUT/$n.$rep.err
    python3 - <<PY
import json
try:
    d = json.load(open("$OUT/$n.$rep.json"))
    big = d.get("roofline_large_batch") or {}
    c5 = d.get("c5_shard_1gpu") or {}
    fi = d.get("from_idle_gpu") or {}
    fl = d.get("roofline_float_pcm") or {}
    print("%-22s rep $rep  C2 busy %.2f us idle %.2f us  big %s ms  c5 %s ms  f32 %s ms  parity %s/%s" % (
        "$n", d["roofline"]["avg_launch_ms"] * 1e3, fi.get("avg_launch_ms", 0) * 1e3, big.get("avg_launch_ms"), c5.get("avg_launch_ms"),
        fl.get("avg_launch_ms"), (d.get("parity") or {}).get("max_abs_diff_lsb"), ((c5.get("parity") or {}).get("max_abs_diff_lsb"))))
except Exception as e:
    print("$n", "failed", e)
PY
  done
done 2>&1 | tee $OUT/summary.txt
