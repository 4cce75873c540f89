#!/bin/bash
# Reproduces the rocprofv3 summaries under profiles/ (run on the GPU box: gpurun -- bash tools/profile.sh <tag>).
# Counter passes are separate runs with --kernel-trace only (no API tracing), as the pool requires; every pass must succeed (set -e):
# a failed pass would otherwise leave a stale or missing CSV behind the summary.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
ENVD="python3 $R/tools/prof_env.py 4096 plane"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- $BENCH > $OUT.bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env -- $ENVD > $OUT.env.log 2>&1
# env-step kernel: HBM traffic + SQ counters
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $ENVD > $OUT.pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $ENVD > $OUT.pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $ENVD > $OUT.pmc3.log 2>&1
# update-phase kernels (the headline roofline kernel among them): HBM traffic of every launch of the bench loop
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcb_fetch -- $BENCH > $OUT.pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcb_write -- $BENCH > $OUT.pmc5.log 2>&1
python3 $R/tools/pmc_summary.py $TAG
cp $(ls -t $OUT/bench/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_bench_kernel_stats.csv
cp $(ls -t $OUT/env/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_envdriver_kernel_stats.csv
