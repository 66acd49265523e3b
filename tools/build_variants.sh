#!/bin/bash
# A/B builds of the engine library into pdmp3_amd/variants/ (git-ignored; they travel to the GPU box with gpurun).
#   bash tools/build_variants.sh name1="flags" name2="flags" ...      e.g.  base="" noslp="-fno-slp-vectorize"
# `git:<rev>` as flags builds the csrc/ of that revision instead of the working tree's.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/pdmp3_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
for spec in "$@"; do
  name=${spec%%=*}; extra=${spec#*=}
  src=$ROOT/pdmp3_amd/csrc
  if [[ $extra == git:* ]]; then
    rev=${extra#git:}; extra=""
    src=/tmp/pdmp3_variant_$name/pdmp3_amd/csrc
    rm -rf /tmp/pdmp3_variant_$name; mkdir -p $src /tmp/pdmp3_variant_$name/include
    for f in $(git -C $ROOT ls-tree --name-only $rev pdmp3_amd/csrc/ include/); do git -C $ROOT show $rev:$f > /tmp/pdmp3_variant_$name/$f; done
  fi
  ( cd $src && hipcc $FLAGS $extra -shared -o $ROOT/pdmp3_amd/variants/$name.so engine.hip $( [ -f node.hip ] && echo node.hip ) -ldl ) &
done
wait
ls -la $ROOT/pdmp3_amd/variants/
