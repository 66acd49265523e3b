#!/bin/bash
# PMC passes on the device-Huffman kernels (k_unpack, k_merge) of the bulk pipeline
OUT=gpurun_out/pmc_unpack
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/bulk_bench.py --frames 20000 --threads 2 --reps 1 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*", "p_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(d)):
        k = r["Kernel_Name"].split("(")[0][:12]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in agg:
        if k.startswith("k_"):
            print(d.split("/")[-2], k, len(disp[k]), {c: round(v / len(disp[k])) for c, v in sorted(agg[k].items())})
PY
