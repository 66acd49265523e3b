#!/bin/bash
# Profiles that get committed under profiles/ (run on the GPU box via gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload (C2 launches only)
#   2. PMC passes (separate runs) on the same workload: FETCH_SIZE, WRITE_SIZE, SQ mix
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu --no-e2e --no-from-idle > $OUT/bench_under_rocprof.json 2> $OUT/stats.log; echo "stats rc=$?"
cat $OUT/stats/bench_kernel_stats.csv
# the same from an idle GPU (no larger launches before the timed region): the clocks' share of the headline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_idle -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu --no-e2e --big 0 --shard 0 > $OUT/bench_idle_under_rocprof.json 2> $OUT/stats_idle.log; echo "idle stats rc=$?"
head -3 $OUT/stats_idle/bench_kernel_stats.csv
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-e2e --big 0 --shard 0 > /dev/null 2> $OUT/$name.log; echo "$name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32
# the whole-stream pipeline (device Huffman): kernel durations per 2048-frame window
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bulk_stats -o bulk -- python3 tools/bulk_bench.py --frames 40000 --threads 2 --reps 2 > $OUT/bulk_under_rocprof.json 2> $OUT/bulk_stats.log; echo "bulk stats rc=$?"
cat $OUT/bulk_stats/bulk_kernel_stats.csv | cut -c1-160
# un-profiled lines
timeout 600 python3 bench.py 2> /dev/null > $OUT/bench.json; echo "bench rc=$?"; cat $OUT/bench.json
timeout 600 python3 tools/bulk_bench.py --frames 137813 --threads 2 2> /dev/null | tail -1 > $OUT/bulk_decode.json; cat $OUT/bulk_decode.json
timeout 600 python3 tools/bulk_bench.py --frames 137813 --threads 1,8,16,32 --host-huffman 2> /dev/null | tail -1 > $OUT/bulk_decode_host_huffman.json; cat $OUT/bulk_decode_host_huffman.json
for j in 1 4 6; do timeout 600 python3 tools/bulk_bench.py --c4 $j 2> /dev/null | tail -1; done > $OUT/bulk_c4.json; cat $OUT/bulk_c4.json
# PCM staying in HBM (a device pointer as destination), one stream and the C4 corpus; where the pipeline's time goes
{ timeout 600 python3 tools/bulk_bench.py --frames 137813 --threads 2 --device-out 2> /dev/null | tail -1; for j in 1 2 4; do timeout 600 python3 tools/bulk_bench.py --c4 $j --device-out 2> /dev/null | tail -1; done; } > $OUT/bulk_device_out.json; cat $OUT/bulk_device_out.json
PDMP3_BULK_TRACE=1 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 2 --device-out --reps 1 2>&1 > /dev/null | grep "bulk trace" | tail -2 > $OUT/bulk_trace.txt; cat $OUT/bulk_trace.txt
# the drop-in streaming API against the number of helper threads; a batch's round trip; k_unpack's phases
for t in 0 1 3 7 11; do PDMP3_STREAM_THREADS=$t timeout 300 python3 tools/stream_api_bench.py 2> /dev/null; done > $OUT/stream_api.json; cat $OUT/stream_api.json
gcc -O2 -Iinclude -o /tmp/rtt tools/stream_rtt.c -Lpdmp3_amd -lpdmp3_hip -Wl,-rpath,$PWD/pdmp3_amd && timeout 120 /tmp/rtt > $OUT/stream_rtt.txt; cat $OUT/stream_rtt.txt
PDMP3_HIP_UNPACK_PROF=1 timeout 300 python3 tools/bulk_bench.py --frames 20000 --threads 2 --reps 1 2>&1 > /dev/null | grep "k_unpack prof" | tail -2 > $OUT/unpack_phases.txt; cat $OUT/unpack_phases.txt
timeout 300 python3 tools/phase_profile.py 131072 32 > $OUT/phase_profile.txt 2>&1; timeout 300 python3 tools/phase_profile.py 2048 1 >> $OUT/phase_profile.txt 2>&1
# round 4: the persistent granule kernel against the engine's own choice, its per-turn stamps, the granule kernel's per-wave stamps,
# the driver's short invocation of the bench
timeout 300 python3 tools/gran_profile.py 2048 > $OUT/gran_profile.txt 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null > $OUT/bench_driver_flags.json; cat $OUT/bench_driver_flags.json | cut -c1-400
# round 4, the whole-stream decoder's split scan: end to end with the PCM left in HBM (the pipeline's waits, a line per window),
# its host side alone, the GPU's timeline of three decodes, a soak with short private windows
{ PDMP3_BULK_TRACE=1 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2>&1 | grep -E "split scan|submitter|frames_per_s" | tail -3
  for w in 4096 16384; do echo "window $w:"; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out --window $w 2> /dev/null | tail -1; done
  for t in 4 12; do echo "scanners $t:"; PDMP3_BULK_SCAN_THREADS=$t timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1; done
  echo "pre-pass in one part:"; PDMP3_BULK_PREPASS_THREADS=1 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 8 --device-out 2> /dev/null | tail -1
  echo "pinned / pageable destination (one-thread scan):"; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 --pinned 2> /dev/null | tail -1; timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 6 2> /dev/null | tail -1
  echo "host side alone (tools/split_scan_bench.py):"; timeout 300 python3 tools/split_scan_bench.py --reps 6 --scanners 1,4,8 --windows 1024 2> /dev/null | tail -1
} > $OUT/split_scan_runs.txt 2>&1; cat $OUT/split_scan_runs.txt | cut -c1-300
PDMP3_BULK_TRACE=2 timeout 300 python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 3 --device-out 2>&1 > /dev/null | grep -E "^  ->|submitter:|split scan|pre-pass in" | tail -45 > $OUT/split_scan_trace.txt
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/bulk_tl -o tl -- python3 tools/bulk_bench.py --frames 137813 --threads 4 --reps 4 --device-out > /dev/null 2> $OUT/bulk_tl.log; python3 tools/bulk_timeline.py $OUT/bulk_tl/tl > $OUT/bulk_timeline.txt 2>&1; cat $OUT/bulk_timeline.txt | cut -c1-260
{ PDMP3_BULK_SCAN_THREADS=8 PDMP3_BULK_SUB_FRAMES=64 timeout 900 python3 tools/soak_bulk.py 20; PDMP3_BULK_SCAN_THREADS=4 PDMP3_BULK_PREPASS_THREADS=5 timeout 900 python3 tools/soak_bulk.py 15; timeout 300 python3 tools/soak_device.py 40; } > $OUT/soak_split.txt 2>&1; tail -3 $OUT/soak_split.txt
# the same for a C5-shard-sized launch
pmcb() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pmc_target.py 131072 0 > /dev/null 2> $OUT/$name.log; echo "$name rc=$?"; }
pmcb big_fetch FETCH_SIZE
pmcb big_write WRITE_SIZE
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
summary = {}
for d in sorted(glob.glob(os.path.join(out, "*", "p_counter_collection.csv"))):
    name = d.split("/")[-2]
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(float); disp = set()
    for r in rows:
        # the C2 passes are the granule kernel's launches only; the big_* passes (tools/pmc_target.py) the chunk kernel's
        want = "k_decode<" if name.startswith("big_") else "k_decode_g"
        if want not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    summary[name] = {"dispatches": n, "per_dispatch": {k: v / n for k, v in sorted(agg.items())}}
    # the same pass's kernel trace: the average duration of the counted dispatches (PMC passes run slower than the plain ones)
    try:
        tr = list(csv.DictReader(open(os.path.join(os.path.dirname(d), "p_kernel_trace.csv"))))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr if want in r.get("Kernel_Name", "")]
        if durs: summary[name]["avg_ns_under_pmc"] = sum(durs) / len(durs)
    except Exception:
        pass
# what the matrix pipe really does: SQ_VALU_MFMA_BUSY_CYCLES / (duration x shader clock x SIMDs), the C2 launch
SCLK_HZ, SIMDS = 2.4e9, 1024
if "sq3" in summary and summary["sq3"].get("avg_ns_under_pmc"):
    busy = summary["sq3"]["per_dispatch"].get("SQ_VALU_MFMA_BUSY_CYCLES")
    if busy:
        summary["mfma_busy_frac"] = busy / (summary["sq3"]["avg_ns_under_pmc"] * 1e-9 * SCLK_HZ * SIMDS)
        summary["mfma_busy_frac_how"] = "SQ_VALU_MFMA_BUSY_CYCLES per dispatch / (that pass's average kernel duration x 2.4 GHz x 1024 SIMDs), k_decode_g at C2"
import hashlib
h = hashlib.sha256()
for f in ("pdmp3_amd/csrc/decode_core.h", "pdmp3_amd/csrc/engine.hip"):
    h.update(open(f, "rb").read())
summary["kernel_source_sha16"] = h.hexdigest()[:16]      # bench.py quotes `traffic` only for the kernel it was measured on
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
