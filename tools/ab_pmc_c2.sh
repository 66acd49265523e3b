#!/bin/bash
# VALU count + launch time of the C2 launch for every engine build in pdmp3_amd/variants/
TAG=${1:-abc2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for so in pdmp3_amd/variants/*.so; do
  n=$(basename $so .so)
  export PDMP3_HIP_LIB=$PWD/$so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/$n.sq1 -o p -- python3 tools/pmc_target.py 2048 0 > $OUT/$n.log 2>&1
  python3 tools/occ_test.py 2048 2>/dev/null | sed "s/^/$n /"
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*", "*", "p_counter_collection.csv")) + glob.glob(os.path.join(out, "*", "p_counter_collection.csv"))):
    rows = list(csv.DictReader(open(d)))
    agg = collections.defaultdict(float); disp = set()
    for r in rows:
        if "k_decode" not in r.get("Kernel_Name", ""): continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(1, len(disp))
    print(d.replace(out + "/", "").split("/")[0], " ".join("%s=%.4g" % (k, v / n / 4096) for k, v in sorted(agg.items())), "(per granule)")
PY
