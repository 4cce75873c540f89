# the split weight-gradient launch behind the chains, in the loop (BG_WGRAD_SPLIT=9 against 0), after the tests of the switch
mkdir -p gpurun_out/r06
timeout -k 10 500 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_mlp_split.py -x -q -m gpu > gpurun_out/r06/wgrad_split_tests.log 2>&1; tail -3 gpurun_out/r06/wgrad_split_tests.log | cut -c1-300
for v in "BG_WGRAD_SPLIT=0" "BG_WGRAD_SPLIT=9" "BG_WGRAD_SPLIT=0" "BG_WGRAD_SPLIT=9"; do echo "$v"; env $v timeout -k 10 120 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/wgrad_split_loop.log 2>&1; cat gpurun_out/r06/wgrad_split_loop.log
