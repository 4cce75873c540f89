# throughput of the loop against the number of envs (flat terrain, no timing events): does the planner's CU split hold at every size?
mkdir -p gpurun_out/r06
for n in 1024 2048 4096 8192 16384 32768; do timeout -k 10 200 python tools/loop_time.py 10 3 2 $n 2>&1 | grep "no instr"; done > gpurun_out/r06/throughput_vs_envs.log 2>&1; cat gpurun_out/r06/throughput_vs_envs.log
