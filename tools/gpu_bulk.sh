#!/bin/bash
# GPU box: bulk pipeline tests + end-to-end throughput (C3-sized stream)
TAG=${1:-bulk}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_bulk.py tests/test_gpu_api.py -m gpu -x -q > $OUT/pytest_bulk.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_bulk.log
timeout 600 python3 tools/bulk_bench.py --frames ${FRAMES:-137813} --threads ${THREADS:-1,8,16,32,64} --parse-only 2>&1 | tee $OUT/bulk_parse.json
timeout 600 python3 tools/bulk_bench.py --frames ${FRAMES:-137813} --threads ${THREADS:-1,8,16,32,64} --host-huffman 2>&1 | tee $OUT/bulk_decode_host_huffman.json
timeout 600 python3 tools/bulk_bench.py --frames ${FRAMES:-137813} --threads ${THREADS:-1,8,16,32,64} 2>&1 | tee $OUT/bulk_decode.json
