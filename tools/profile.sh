#!/bin/bash
# Reproduces the rocprofv3 summaries under profiles/ (run on the GPU box: gpurun -- bash tools/profile.sh <tag>).
# Counter passes are separate runs with --kernel-trace only (no API tracing), as the pool requires.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT.bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env -- python3 $R/tools/prof_env.py 4096 plane > $OUT.env.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_env.py 4096 plane > $OUT.pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_env.py 4096 plane > $OUT.pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_env.py 4096 plane > $OUT.pmc3.log 2>&1 || true
ls -R $OUT | head -40
