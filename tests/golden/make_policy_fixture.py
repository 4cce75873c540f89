"""Outputs of the reference's OWN deploy-side code, as numbers: tests/golden/deploy_policy.npz.

Runs /root/reference/deploy/utils/policy.py:Policy (imported, read-only; numpy + torch only) with /root/reference/deploy/configs/T1.yaml and the trained
TorchScript actor /root/reference/deploy/models/T1.pt: 8 episodes x 12 consecutive `inference()` calls on seeded robot states (the class is stateful: smoothed
commands, the previous actions, the gait switch of policy.py:42-45), among them zero-command episodes (gait gate closed) and commands that ramp up from zero
(the smoothing of policy.py:39-40).  Stored per call: the inputs, the state the class carried INTO the observation (smoothed commands, gait frequency and
process), the 47 observations (policy.py:47-62), the raw network output, the clipped actions (policy.py:64-69) and the 23 joint targets (policy.py:70-71).
Only numbers are stored; neither the code nor the TorchScript archive is copied.

Run in the build container:  python tests/golden/make_policy_fixture.py
"""
import importlib.util
import os

import numpy as np
import torch
import yaml

REF = "/root/reference/deploy"
spec = importlib.util.spec_from_file_location("ref_deploy_policy", os.path.join(REF, "utils", "policy.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
cfg = yaml.safe_load(open(os.path.join(REF, "configs", "T1.yaml")))
cfg["policy"]["policy_path"] = os.path.join(REF, "models", "T1.pt")

rng = np.random.default_rng(20251005)
EPISODES, STEPS = 8, 12
keys = ("time", "dof_pos", "dof_vel", "base_ang_vel", "projected_gravity", "command", "smoothed_commands", "gait_frequency", "gait_process", "obs", "raw_actions",
        "actions", "dof_targets")
rec = {k: [] for k in keys}
for ep in range(EPISODES):
    p = mod.Policy(cfg)
    default = np.array(cfg["common"]["default_qpos"], dtype=np.float32)
    cmd = np.zeros(3, np.float32) if ep in (0, 5) else rng.uniform(-1, 1, 3).astype(np.float32)   # two zero-command episodes: the gait gate stays closed
    t = float(rng.uniform(0, 5))
    for k in range(STEPS):
        if ep == 5 and k == 6:
            cmd = np.array([0.6, 0.0, -0.3], np.float32)   # ... and one that starts to walk in the middle: the commands ramp up by 0.02 per call
        dof_pos = (default + rng.normal(size=23) * 0.15).astype(np.float32)
        dof_vel = (rng.normal(size=23) * 1.5).astype(np.float32)
        w = (rng.normal(size=3) * 0.5).astype(np.float32)
        g = rng.normal(size=3) * 0.15 + np.array([0, 0, -1.0]); g = (g / np.linalg.norm(g)).astype(np.float32)
        targets = p.inference(t, dof_pos, dof_vel, w, g, float(cmd[0]), float(cmd[1]), float(cmd[2]))
        with torch.no_grad():
            raw = p.policy(torch.from_numpy(p.obs).unsqueeze(0)).numpy()[0]   # the same module on the same observation: the output before the clip
        for name, v in (("time", t), ("dof_pos", dof_pos), ("dof_vel", dof_vel), ("base_ang_vel", w), ("projected_gravity", g), ("command", cmd.copy()),
                        ("smoothed_commands", p.smoothed_commands.copy()), ("gait_frequency", float(p.gait_frequency)), ("gait_process", float(p.gait_process)),
                        ("obs", p.obs.copy()), ("raw_actions", raw.copy()), ("actions", p.actions.copy()), ("dof_targets", np.array(targets).copy())):
            rec[name].append(v)
        t += p.get_policy_interval()
out = {k: np.array(v, dtype=np.float32 if k != "time" else np.float64).reshape(EPISODES, STEPS, *np.shape(v[0])) for k, v in rec.items()}
out["default_qpos"] = np.array(cfg["common"]["default_qpos"], dtype=np.float32)
out["policy_interval"] = np.float64(cfg["common"]["dt"] * cfg["policy"]["control"]["decimation"])
np.savez_compressed("tests/golden/deploy_policy.npz", **out)
print({k: v.shape for k, v in out.items()})
print("gate closed in", int((out["gait_frequency"] == 0).sum()), "of", EPISODES * STEPS, "calls; |raw| max", float(np.abs(out["raw_actions"]).max()))
