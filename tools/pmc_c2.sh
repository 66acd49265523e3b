#!/bin/bash
# PMC instruction mix of the C2 launch (2048 frames) for the chain modes.  Usage: gpurun -- 'bash tools/pmc_c2.sh TAG'
TAG=${1:-pmcc2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for m in 0 1; do
  export PDMP3_HIP_CHAIN=$m
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/m$m.sq1 -o p -- python3 tools/pmc_target.py 2048 0 > $OUT/m$m.sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/m$m.sq3 -o p -- python3 tools/pmc_target.py 2048 0 > $OUT/m$m.sq3.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*", "*", "p_counter_collection.csv")) + glob.glob(os.path.join(out, "*", "p_counter_collection.csv"))):
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(float); disp = set(); names=set()
    for r in rows:
        if "k_decode" not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"]); names.add(r["Kernel_Name"][:24])
    n = max(1, len(disp))
    print(d.replace(out + "/", "").split("/")[0], sorted(names), " ".join("%s=%.4g" % (k, v / n) for k, v in sorted(agg.items())))
PY
