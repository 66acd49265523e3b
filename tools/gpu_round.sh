#!/bin/bash
# One GPU-box round: smoke, GPU parity tests, bench, rocprof kernel trace.
# Usage (from the repo root): gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/rocminfo.txt 2>&1
nproc > $OUT/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/nproc.txt
echo "== smoke" | tee $OUT/smoke.log
timeout 600 python3 __graft_entry__.py smoke >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== chunk sweep"
for c in 2 4 8 16; do timeout 120 python3 bench.py --chunk $c --steps 100 --warmup 10 --no-cpu --big 131072 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('chunk', '$c', 'fps', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'big', d.get('roofline_large_batch'))
"; done 2>&1 | tee $OUT/chunk_sweep.txt
echo "== rocprof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu --big 131072 > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do cat $f; done
