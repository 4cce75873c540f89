# one-stream mini-epoch (both networks' chains in one launch each, round 6) against the two-stream form: tests, then the loop
mkdir -p gpurun_out/r06
timeout -k 10 600 python -m pytest tests/test_gpu_ppo.py -x -q -m gpu > gpurun_out/r06/one_stream_tests.log 2>&1; tail -4 gpurun_out/r06/one_stream_tests.log
for v in "BG_ONE_STREAM=1" "BG_ONE_STREAM=0" "BG_ONE_STREAM=1" "BG_ONE_STREAM=0"; do echo "$v"; env $v timeout -k 10 120 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/one_stream_loop.log 2>&1; cat gpurun_out/r06/one_stream_loop.log
