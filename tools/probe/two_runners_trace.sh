set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tr2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/probe/two_runners.py repeat 16384 3 > $OUT.log 2>&1
grep "runner" $OUT.log | cut -c1-60
python3 $R/tools/probe/two_runners_trace.py $(ls -t $OUT/*/*kernel_trace.csv | head -1) > $R/gpurun_out/two_runners_trace.txt
cat $R/gpurun_out/two_runners_trace.txt
