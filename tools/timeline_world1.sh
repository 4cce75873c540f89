set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tl_w1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BG_DIST_FORCE=1
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT.log 2>&1
python3 $R/tools/timeline.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) > $R/gpurun_out/timeline_w1.txt
cat $R/gpurun_out/timeline_w1.txt
