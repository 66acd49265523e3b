#!/bin/bash
# The granule kernel's chain along the XCDs (PDMP3_HIP_XCD_CHAIN=1: places p, p + 1 on one XCD, the state between them in plain
# stores that stay in its L2) against the chain by blockIdx: HBM traffic of a C2 launch (FETCH_SIZE / WRITE_SIZE), also with
# workgroups of 8 waves (twice the hand-overs between workgroups: what a hand-over costs in either counter).
OUT=gpurun_out/xcd_chain
mkdir -p $OUT
export TMPDIR=/tmp
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-e2e --big 0 --shard 0 > /dev/null 2> $OUT/$name.log; }
for m in 0 1; do for w in 16 8; do
  export PDMP3_HIP_XCD_CHAIN=$m PDMP3_HIP_GRAN_W=$w
  pmc fetch_c${m}_w$w FETCH_SIZE; pmc write_c${m}_w$w WRITE_SIZE; pmc tcc_c${m}_w$w TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum
done; done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*_c*", "p_counter_collection.csv"))):
    name = d.split("/")[-2]
    agg = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(d)):
        if "k_decode_g" not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    print(name, {k: round(v / n) for k, v in agg.items()}, "dispatches", n)
PY
