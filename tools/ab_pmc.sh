#!/bin/bash
# PMC instruction mix of the 131072-frame launch for every engine build in pdmp3_amd/variants/ (one box).
# Usage: gpurun --timeout 900 -- 'bash tools/ab_pmc.sh TAG'
TAG=${1:-abpmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for so in pdmp3_amd/variants/*.so; do
  n=$(basename $so .so)
  export PDMP3_HIP_LIB=$PWD/$so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/$n.sq1 -o p -- python3 tools/pmc_target.py 131072 0 > $OUT/$n.sq1.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$n.sq2 -o p -- python3 tools/pmc_target.py 131072 0 > $OUT/$n.sq2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/$n.sq3 -o p -- python3 tools/pmc_target.py 131072 0 > $OUT/$n.sq3.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*", "p_counter_collection.csv")) + glob.glob(os.path.join(out, "*", "*", "p_counter_collection.csv"))):
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(float); disp = set()
    for r in rows:
        if "k_decode" not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    print(d.replace(out + "/", "").split("/")[0], " ".join("%s=%.4g" % (k, v / n) for k, v in sorted(agg.items())))
PY
