mkdir -p gpurun_out/r06
timeout -k 10 300 python -m pytest tests/test_gpu_mlp_chain_split.py -x -q 2>&1 | tail -8 > gpurun_out/r06/chain_split_test_v3.log; cat gpurun_out/r06/chain_split_test_v3.log
timeout -k 10 200 python tools/chain_split_probe.py 50 > gpurun_out/r06/chain_split_probe_v3.log 2>&1; grep -v "^{" gpurun_out/r06/chain_split_probe_v3.log | tail -12
for v in stamps nodma nofin nosplit nostore bare; do echo "== $v"; BG_LIB=$GRAFT_REPO_ROOT/tools/probe/libbg_split_$v.so timeout -k 10 200 python tools/chain_split_stamps.py > gpurun_out/r06/split_stamps_v3_$v.log 2>&1; grep -A2 "alone" gpurun_out/r06/split_stamps_v3_$v.log; done
