#!/bin/bash
# quick GPU iteration: parity tests + phase profile + short bench
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python3 tools/phase_profile.py 131072 32 2>&1 | tee $OUT/phase_131072_32.txt
timeout 300 python3 tools/phase_profile.py 2048 2 2>&1 | tee $OUT/phase_2048_2.txt
timeout 300 python3 bench.py --steps 100 --warmup 10 --no-cpu 2>/dev/null | tee $OUT/bench.json
