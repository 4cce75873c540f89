mkdir -p gpurun_out/r06
timeout -k 10 600 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_mlp_chain_split.py tests/test_gpu_head.py -x -q 2>&1 | tail -15 > gpurun_out/r06/int1_tests.log; cat gpurun_out/r06/int1_tests.log
for v in 1 0 1 0; do echo "BG_CHAIN_SPLIT=$v"; BG_CHAIN_SPLIT=$v timeout -k 10 200 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/int1_loop.log 2>&1; cat gpurun_out/r06/int1_loop.log
