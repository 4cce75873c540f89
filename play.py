"""python play.py --task=T1 --checkpoint=-1   (reference play.py:1-6): deterministic rollout of a trained policy.

The reference records a camera video; a headless MI355X node has no renderer, so `BG_PLAY_STEPS` / `BG_PLAY_RECORD` optionally
bound the rollout and dump env 0's trajectory to an .npz instead."""
import os

from booster_gym_amd.utils.runner import Runner

if __name__ == "__main__":
    runner = Runner(test=True)
    steps = os.environ.get("BG_PLAY_STEPS")
    runner.play(max_steps=int(steps) if steps else None, record_path=os.environ.get("BG_PLAY_RECORD"))
