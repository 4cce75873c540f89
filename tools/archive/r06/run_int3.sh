mkdir -p gpurun_out/r06
timeout -k 10 600 python -m pytest tests/test_gpu_mlp_chain_split_bwd.py tests/test_gpu_ppo.py -x -q 2>&1 | tail -8 > gpurun_out/r06/int3_tests.log; cat gpurun_out/r06/int3_tests.log
timeout -k 10 200 python tools/chain_split_bwd_probe.py 50 > gpurun_out/r06/bwd_probe_v2.log 2>&1; grep -v "^{" gpurun_out/r06/bwd_probe_v2.log | tail -16
for v in "BG_CHAIN_SPLIT_BWD=1" "BG_CHAIN_SPLIT_BWD=0" "BG_CHAIN_SPLIT_BWD=1 BG_SPLIT_BWD_CHAIN_CUS=0" "BG_CHAIN_SPLIT_BWD=1" "BG_CHAIN_SPLIT_BWD=0"; do echo "$v"; env $v timeout -k 10 200 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/int3_loop.log 2>&1; cat gpurun_out/r06/int3_loop.log
