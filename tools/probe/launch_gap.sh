set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/lg
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/probe/launch_gap.py > $OUT.log 2>&1
python3 $R/tools/probe/launch_gap.py --read $(ls -t $OUT/*/*kernel_trace.csv | head -1) > $R/gpurun_out/launch_gap.txt
cat $R/gpurun_out/launch_gap.txt
