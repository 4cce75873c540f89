#!/bin/bash
# One mini-epoch of the bench loop, kernel by kernel: rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2` -> tools/timeline.py (an unarmed iteration: idx -3).
#   gpurun -- bash tools/timeline_once.sh <tag> [ENV=VALUE ...]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT.log 2>&1
python3 $R/tools/timeline.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) > $R/gpurun_out/timeline_$TAG.txt
cat $R/gpurun_out/timeline_$TAG.txt
