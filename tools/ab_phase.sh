#!/bin/bash
# Phase profile (k_decode_prof) of every engine build in pdmp3_amd/variants/ on ONE box.
# Usage: gpurun --timeout 900 -- 'bash tools/ab_phase.sh TAG [n_frames chunk]'
TAG=${1:-abp}
N=${2:-131072}
CH=${3:-32}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for so in pdmp3_amd/variants/*.so; do
  n=$(basename $so .so)
  echo "== $n"
  PDMP3_HIP_LIB=$PWD/$so timeout 300 python3 tools/phase_profile.py $N $CH 2>/dev/null | tee $OUT/$n.phase.txt | head -12
done
