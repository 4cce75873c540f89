# CU shares of the two split chains' launch pairs in the loop (round 6): planner's choice against fixed splits, incl. both launches on all CUs
mkdir -p gpurun_out/r06
for v in "BG_X=0" "BG_BWD_CHAIN_CUS=256,256" "BG_BWD_CHAIN_CUS=176,80" "BG_BWD_CHAIN_CUS=168,88" "BG_FWD_CHAIN_CUS=256,256" "BG_FWD_CHAIN_CUS=168,88" "BG_FWD_CHAIN_CUS=256,256 BG_BWD_CHAIN_CUS=256,256" "BG_X=0"; do echo "$v"; env $v timeout -k 10 120 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/cu_sweep.log 2>&1; cat gpurun_out/r06/cu_sweep.log
