#!/bin/bash
# Long soaks of the final tree on the GPU box (profiles/r06_soak_final.txt):  gpurun --timeout 2400 -- 'bash tools/gpu_soak.sh'
OUT=gpurun_out/soak_final
mkdir -p $OUT
{ timeout 500 python3 tools/soak_stream.py 120 6
  PDMP3_STREAM_SPIN=0 PDMP3_STREAM_THREADS=7 timeout 300 python3 tools/soak_stream.py 60 8
  timeout 900 python3 tools/soak_chain.py 150
  timeout 900 python3 tools/soak_bulk.py 60
  PDMP3_BULK_SCAN_THREADS=8 PDMP3_BULK_SUB_FRAMES=64 timeout 900 python3 tools/soak_bulk.py 40
  timeout 500 python3 tools/soak_device.py 240
  timeout 500 python3 tests/fuzz_gpu.py 180 1
  timeout 500 python3 tests/fuzz_gpu.py 180 2 corrupt
} 2>&1 | grep -v amdgpu.ids | tee $OUT/soak.txt
