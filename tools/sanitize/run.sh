#!/bin/bash
# Sanitizer runs of the host library's thread paths on the CPU build (no GPU: parse-only handles; the engine library is
# only linked).  ThreadSanitizer, then AddressSanitizer + UBSan, over (a) three handles reading concurrently through
# pdmp3_feed / pdmp3_read with the helper pool, (b) the whole-stream parser on 2 / 4 / 6 pool threads.
#   bash tools/sanitize/run.sh file.mp3      (e.g. python -m pdmp3_amd.packer c3 /tmp/c3; any stream of a few thousand frames)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/pdmp3_sanitize
mkdir -p $OUT
HOST=$ROOT/pdmp3_amd/host
SRCS="$HOST/huffman_lut.c $HOST/frame_parse.c $HOST/stream_api.c $HOST/cpus.c $HOST/bulk.c $HOST/split_scan.c $HOST/bulk_api.c $HOST/corpus.c $HOST/wav_cli.c"
for san in thread address,undefined; do
  for t in stream_threads bulk_threads split_scan; do
    gcc -O1 -g -fsanitize=$san -I$ROOT/include -I$ROOT/pdmp3_amd/csrc -o $OUT/$t $ROOT/tools/sanitize/$t.c $SRCS \
        -L$ROOT/pdmp3_amd -lpdmp3_hip -lpthread -Wl,-rpath,$ROOT/pdmp3_amd -w     # (warnings off, errors shown: a failed build stops the script with its message)
    echo "== $san $t"
    ASAN_OPTIONS=detect_leaks=0 $OUT/$t "$1" 2>&1 | tail -4
  done
done
