mkdir -p gpurun_out/r06
for v in "BG_BWD_CHAIN_CUS=160,96" "BG_BWD_CHAIN_CUS=176,80" "BG_BWD_CHAIN_CUS=152,104" "BG_BWD_CHAIN_CUS=144,112" "BG_BWD_CHAIN_CUS=168,88" "BG_BWD_CHAIN_CUS=128,128"; do echo "$v"; env $v timeout -k 10 200 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/int4_loop.log 2>&1; cat gpurun_out/r06/int4_loop.log
