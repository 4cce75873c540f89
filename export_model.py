"""python export_model.py --task=T1 [--checkpoint path|-1]: TorchScript export of the actor (reference export_model.py:8-30) so that
the reference's deployment code (deploy/utils/policy.py:9) can load what this framework trains."""
import argparse
import glob
import os

import torch

from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.model import ActorCritic

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", required=True, type=str)
    ap.add_argument("--checkpoint", type=str, default="-1")
    args = ap.parse_args()
    cfg = load_cfg(args.task)
    ck = args.checkpoint
    if ck in ("-1", None):
        ck = sorted(glob.glob(os.path.join("logs", "**/*.pth"), recursive=True), key=os.path.getmtime)[-1]
    print("Loading model from {}".format(ck))
    model = ActorCritic(cfg["env"]["num_actions"], cfg["env"]["num_observations"], cfg["env"]["num_privileged_obs"])
    model.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True)["model"])
    os.makedirs("deploy/models", exist_ok=True)
    out = os.path.join("deploy", "models", f"{args.task}.pt")
    torch.jit.script(model.actor).save(out)
    print("Exported actor to {}".format(out))
