timeout 900 python -m pytest tests/test_gpu_bulk.py tests/test_gpu_corpus.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rows_stats -o b -- python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 4 --device-out > /dev/null 2>&1
grep -E "k_rows|k_unpack|k_merge|k_decode_g" gpurun_out/rows_stats/b_kernel_stats.csv | cut -c1-140
for i in 1 2 3; do timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2>&1 | tail -1 | cut -c185-260; done
