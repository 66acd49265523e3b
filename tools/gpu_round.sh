#!/bin/bash
# One GPU-box round: smoke, GPU parity tests, bench.
# Usage (from the repo root): gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke"
timeout 600 python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
tail -3 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 2400 python3 -m pytest tests -m gpu -x -q -s --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
