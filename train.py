"""python train.py --task=T1 [--num_envs N --seed S --max_iterations K --sim_device cuda:0 --rl_device cuda:0 --checkpoint P]

Entry point with the reference's surface (reference train.py:1-6).  Multi-GPU: `torchrun --nproc-per-node N train.py --task=T1`
(one rank per GPU, environments sharded, gradients all-reduced over RCCL)."""
from booster_gym_amd.utils.runner import Runner

if __name__ == "__main__":
    runner = Runner(test=False)
    runner.train()
    runner.dp.shutdown()
