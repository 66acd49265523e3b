timeout 900 python -m pytest tests/test_gpu_bulk.py -x -q 2>&1 | tail -3
for i in 1 2; do
PDMP3_BULK_TRACE=1 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2>&1 | grep -E "split scan|frames_per_s" | tail -2 | cut -c1-600
done
for g in 3 8; do echo "== gather helpers $g"; PDMP3_BULK_GATHER_THREADS=$g timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2>&1 | grep -E "frames_per_s" | tail -1 | cut -c150-420; done
echo "== pinned default"; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 --pinned 2>&1 | grep -E "frames_per_s" | tail -1 | cut -c150-420
echo "== pageable default"; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 2>&1 | grep -E "frames_per_s" | tail -1 | cut -c150-420
PDMP3_BULK_TRACE=2 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 3 --device-out 2>&1 | grep -E "^  ->|submitter:|split scan" | tail -42
