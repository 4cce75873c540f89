mkdir -p gpurun_out/r06
timeout -k 10 300 python -m pytest tests/test_gpu_mlp_chain_split_bwd.py -x -q 2>&1 | tail -4 > gpurun_out/r06/bwd_test_v3.log; cat gpurun_out/r06/bwd_test_v3.log
for v in stamps nostore; do echo "== $v"; BG_LIB=$GRAFT_REPO_ROOT/tools/probe/libbg_bwd_$v.so timeout -k 10 200 python tools/chain_split_bwd_stamps.py $1 $2 > gpurun_out/r06/bwd_stamps_$v.log 2>&1; grep -A2 "alone" gpurun_out/r06/bwd_stamps_$v.log; done
