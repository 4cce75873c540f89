# steps per group of the first mini-epoch's forward passes during the rollout (BG_ROLLOUT_FORWARD_GROUP), in the loop
mkdir -p gpurun_out/r06
for g in 2 1 3 4 6 2; do echo "BG_ROLLOUT_FORWARD_GROUP=$g"; BG_ROLLOUT_FORWARD_GROUP=$g timeout -k 10 120 python tools/loop_time.py 20 5 2 2>&1 | grep "no instr"; done > gpurun_out/r06/rollout_group.log 2>&1; cat gpurun_out/r06/rollout_group.log
