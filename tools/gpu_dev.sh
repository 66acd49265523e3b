#!/bin/bash
# Kernel development round on the GPU box: parity tests of the transform kernel, bench line, phase profile.
# Usage (from the repo root): gpurun --timeout 900 -- 'bash tools/gpu_dev.sh TAG [quick]'
TAG=${1:-dev}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" = "quick" ]; then
  timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
else
  timeout 800 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
fi
tail -4 $OUT/pytest.log
timeout 300 python3 bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("C2 value %.1f M  launch %.2f us  frac %.4f | big %.3f ms %.1f M frac %.4f | e2e %.2f M" % (
    d["value"] / 1e6, d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["frac"],
    d["roofline_large_batch"]["avg_launch_ms"], d["roofline_large_batch"]["frames_per_s"] / 1e6, d["roofline_large_batch"]["frac"],
    d.get("end_to_end", {}).get("frames_per_s", 0) / 1e6))
PY
timeout 200 python3 tools/phase_profile.py 131072 32 > $OUT/phase.txt 2>&1
timeout 200 python3 tools/phase_profile.py 2048 1 >> $OUT/phase.txt 2>&1
cat $OUT/phase.txt
