mkdir -p gpurun_out/r06
timeout -k 10 900 python -m pytest tests/test_gpu_mlp_chain_split.py tests/test_gpu_mlp_chain_split_bwd.py tests/test_gpu_ppo.py tests/test_gpu_mlp_split.py -x -q -s 2>&1 | grep -v "^split chain\|^split backward" | tail -12 > gpurun_out/r06/int5_tests.log; cat gpurun_out/r06/int5_tests.log
for v in "BG_CHAIN_ALTERNATE=1" "BG_CHAIN_ALTERNATE=0" "BG_CHAIN_ALTERNATE=1"; do echo "$v"; env $v timeout -k 10 200 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/int5_loop.log 2>&1; cat gpurun_out/r06/int5_loop.log
