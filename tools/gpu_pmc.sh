#!/bin/bash
# PMC passes (each its own run; --pmc is never combined with other trace domains except --kernel-trace)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT 2>/dev/null || true
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pmc_target.py 131072 32 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC
run grbm GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("OUTDIR", "") or "gpurun_out/%s" % os.environ.get("TAG", "pmc")
PY
for f in $(find $OUT -name "*counter_collection.csv"); do
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "")[:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
seen = set()
for r in rows:
    key = (r.get("Kernel_Name", "")[:40], r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key); cnt[key[0]] += 1
for k in agg:
    if "k_decode" not in k: continue
    print(sys.argv[1].split("/")[-3], k, "dispatches", cnt[k])
    for c, v in sorted(agg[k].items()):
        print("   %-28s %16.0f per dispatch %14.1f" % (c, v, v / max(1, cnt[k])))
PY
done 2>&1 | tee $OUT/summary.txt
