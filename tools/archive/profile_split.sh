#!/bin/bash
# rocprofv3 kernel summary of the bench loop in the opt-in split-bf16 mode (BG_GEMM_SPLIT=9 and 6): gpurun -- bash tools/profile_split.sh
# The variable is exported BEFORE rocprofv3 starts (no env / bash -c hop between the profiler and python).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for s in 9 6; do
  OUT=$R/gpurun_out/prof_split$s
  mkdir -p $OUT
  export BG_GEMM_SPLIT=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT.bench.log 2>&1
  cp $(ls -t $OUT/bench/*/*kernel_stats.csv | head -1) $R/gpurun_out/r02_bench_split${s}_kernel_stats.csv
done
head -14 $R/gpurun_out/r02_bench_split9_kernel_stats.csv | cut -c1-150
