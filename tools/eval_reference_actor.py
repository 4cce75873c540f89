"""How well does a policy trained by the REFERENCE (PhysX TGS, Isaac Gym) survive and track in this build's simulator?

The reference ships one trained actor (deploy/models/T1.pt; its weights only are kept as tests/golden/t1_actor.npz).  Its weights encode the
PhysX dynamics it was trained on, so its behaviour here is the one quantitative statement about PhysX-versus-this-contact-model that can be made
offline (MuJoCo and Isaac Gym are absent).  The actor drives `num_envs` robots for a full episode (1,500 env steps = 30 s) under the SHIPPED
envs/T1.yaml -- observation noise, domain randomisation, actuation latency, kicks and pushes all on, commands resampled as in training --
exactly as play_mujoco.py:734-756 / deploy/utils/policy.py:47-62 would feed it (47 observations -> 12 actions, deterministic mean).

Reported per scenario (terrain x asset.self_collisions [x contact overrides]):
  fall_rate_first_episode   fraction of robots whose FIRST episode ends in a termination (height / velocity / contact), not the time-out
  falls_per_robot_minute    terminations over the whole run, per robot and simulated minute (robots are reset and go on, as in training)
  mean_first_episode_length mean length in env steps of every robot's first episode (1,501 = all of them ran to the time-out); fell_within_steps = the
                            cumulative fall fraction over episode time; falls_by_command = what the fallen robots had been told to do
  tracking_rmse             per axis: sqrt(mean((command - filtered velocity)^2)) over robots given a moving command (|cmd| > 0), after the first
                            second of each episode; `filtered_*` is what the tracking rewards see (t1.py:610-620); `tracking_rmse_still` for cmd = 0
  reward_terms              mean of each scaled reward term per env step (extras["rew_terms"], t1.py:566-570) and of the total

    python tools/eval_reference_actor.py [num_envs=4096] [steps=1600] [--sweep] [--checkpoint logs/.../model.pth] -> gpurun_out/reference_actor_eval.json
(--checkpoint: the same protocol for an actor trained by THIS build -> gpurun_out/own_actor_eval.json: the yardstick for the reference actor's numbers)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from booster_gym_amd.envs import T1
from booster_gym_amd.utils.config import load_cfg


ACTOR_CHECKPOINT = None  # --checkpoint <.pth of this build's Runner>: evaluate that actor instead (how a policy TRAINED HERE fares under the same protocol)


def load_actor(dev):
    if ACTOR_CHECKPOINT:
        sd = torch.load(ACTOR_CHECKPOINT, map_location=dev, weights_only=True)["model"]
        W = {k[len("actor."):]: v.float() for k, v in sd.items() if k.startswith("actor.")}
        layers = [(W[f"{i}.weight"].to(dev), W[f"{i}.bias"].to(dev)) for i in (0, 2, 4, 6)]
    else:
        W = np.load(os.path.join(ROOT, "tests", "golden", "t1_actor.npz"))
        layers = [(torch.tensor(W[f"{i}.weight"], device=dev), torch.tensor(W[f"{i}.bias"], device=dev)) for i in (0, 2, 4, 6)]

    def actor(x):
        for k, (w, b) in enumerate(layers):
            x = torch.addmm(b, x, w.T)
            if k < 3:
                x = torch.nn.functional.elu(x)
        return x

    return actor


def evaluate(n, steps, overrides, seed=42):
    cfg = load_cfg("T1", dict({"env.num_envs": n, "basic.seed": seed}, **overrides))
    env = T1(cfg)
    dev = env.device
    actor = load_actor(dev)
    obs, _ = env.reset()
    first_done = torch.zeros(n, dtype=torch.bool, device=dev)   # the first episode of this robot has ended
    first_fell = torch.zeros(n, dtype=torch.bool, device=dev)
    first_len = torch.zeros(n, dtype=torch.int32, device=dev)    # length of the first episode
    fall_cmd = torch.zeros(n, 3, device=dev)                     # the command a robot had when its first episode ended in a fall
    cmd_prev = env.commands
    falls = torch.zeros((), dtype=torch.float64, device=dev)
    sq = torch.zeros(3, dtype=torch.float64, device=dev); cnt = torch.zeros(3, dtype=torch.float64, device=dev)
    sq0 = torch.zeros(3, dtype=torch.float64, device=dev); cnt0 = torch.zeros(3, dtype=torch.float64, device=dev)
    terms = {k: torch.zeros((), dtype=torch.float64, device=dev) for k in env.reward_names}
    rew_sum = torch.zeros((), dtype=torch.float64, device=dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(steps):
            obs, rew, done, extras = env.step(actor(obs))
            fell = done & ~extras["time_outs"]
            falls += fell.sum()
            newly = fell & ~first_done
            first_fell |= newly
            fall_cmd[newly] = cmd_prev[newly]
            first_len += (~first_done).int()
            first_done |= done
            cmd_prev = env.commands  # (a reset env already holds its next command)
            rew_sum += rew.double().sum()
            for k in terms:
                terms[k] += extras["rew_terms"][k].double().sum()
            if s % 5 == 0:  # tracking error, every fifth step
                cmd = cmd_prev
                v = torch.cat((env.get_field("filtered_lin_vel")[:, :2], env.get_field("filtered_ang_vel")[:, 2:3]), dim=1)
                settled = (env.episode_length_buf > 50)[:, None]
                err2 = (cmd - v).double() ** 2
                moving = (cmd.abs() > 1e-6) & settled
                still = (cmd.abs() <= 1e-6) & settled
                sq += (err2 * moving).sum(0); cnt += moving.sum(0)
                sq0 += (err2 * still).sum(0); cnt0 += still.sum(0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = env.episode_stats(reset=False).double().cpu().numpy()  # finished episodes, sum of lengths, sum of reward, 26 term sums, non-finite resets
    ax = ("lin_vel_x", "lin_vel_y", "ang_vel_yaw")
    out = {"overrides": overrides, "num_envs": n, "steps": steps, "wall_s": wall,
           "fall_rate_first_episode": float(first_fell.double().mean()),
           "fell_within_steps": {str(k): float((first_fell & (first_len <= k)).double().mean()) for k in (100, 300, 500, 1000, 1500)},
           "mean_first_episode_length": float(first_len.double().mean()),
           "falls_by_command": {"standing (cmd = 0)": int((first_fell & (fall_cmd.abs().sum(1) == 0)).sum()),
                                "|vx| > 0.5": int((first_fell & (fall_cmd[:, 0].abs() > 0.5)).sum()), "|vy| > 0.5": int((first_fell & (fall_cmd[:, 1].abs() > 0.5)).sum()),
                                "|yaw| > 0.5": int((first_fell & (fall_cmd[:, 2].abs() > 0.5)).sum()), "all": int(first_fell.sum())},
           "falls_total": float(falls), "falls_per_robot_minute": float(falls) / (n * steps * env.dt / 60.0),
           "finished_episodes": float(st[0]), "mean_episode_length": float(st[1] / max(st[0], 1.0)),
           "first_episode_ran_to_the_time_out": float((first_done & ~first_fell).double().mean()),
           "tracking_rmse": {a: float(torch.sqrt(sq[i] / cnt[i].clamp(min=1))) for i, a in enumerate(ax)},
           "tracking_rmse_still": {a: float(torch.sqrt(sq0[i] / cnt0[i].clamp(min=1))) for i, a in enumerate(ax)},
           "mean_reward_per_step": float(rew_sum) / (n * steps),
           "reward_terms": {k: float(v) / (n * steps) for k, v in terms.items()},
           "nonfinite_resets": float(st[-1])}
    del env
    return out


def main():
    global ACTOR_CHECKPOINT
    if "--checkpoint" in sys.argv:
        k = sys.argv.index("--checkpoint")
        ACTOR_CHECKPOINT = sys.argv[k + 1]
        del sys.argv[k : k + 2]
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 4096
    steps = int(args[1]) if len(args) > 1 else 1600  # one full episode (1,500 steps) and its time-out
    res = {"what": __doc__.split("\n\n")[0],
           "actor": f"{ACTOR_CHECKPOINT} (an actor trained by this build)" if ACTOR_CHECKPOINT else "tests/golden/t1_actor.npz (weights of the reference's deploy/models/T1.pt)",
           "config": "envs/T1.yaml as shipped",
           "scenarios": {}}
    for terrain in ("plane", "trimesh"):
        for sc in (0, 1):
            key = f"{terrain}/self_collisions={sc}"
            res["scenarios"][key] = evaluate(n, steps, {"terrain.type": terrain, "asset.self_collisions": sc})
            r = res["scenarios"][key]
            print(key, "fall rate (first episode)", round(r["fall_rate_first_episode"], 4), "mean episode length", round(r["mean_first_episode_length"], 1),
                  "rmse", {k: round(v, 3) for k, v in r["tracking_rmse"].items()}, flush=True)
    # the same actor without the perturbations, to separate "the contact model differs" from "kicks, pushes and noise are hard"
    quiet = {"terrain.type": "plane", "noise.gravity": None, "noise.ang_vel": None, "noise.dof_pos": None, "noise.dof_vel": None,
             "randomization.kick_lin_vel": None, "randomization.kick_ang_vel": None, "randomization.push_force": None, "randomization.push_torque": None}
    res["scenarios"]["plane/no_noise_no_kicks_no_pushes"] = evaluate(n, steps, quiet)
    print("quiet", res["scenarios"]["plane/no_noise_no_kicks_no_pushes"]["fall_rate_first_episode"], flush=True)
    if "--sweep" in sys.argv:
        # which contact parameter moves the fall rate: one at a time around the defaults (contact.* of T1.yaml), plane, shipped perturbations
        sweep = {}
        for name, values in (("contact.stiffness", (1.0e4, 2.0e4, 8.0e4, 1.6e5)), ("contact.damping", (150.0, 300.0, 1200.0, 2400.0)),
                             ("contact.friction_viscosity", (2.5e3, 5.0e3, 2.0e4, 4.0e4))):
            for v in values:
                r = evaluate(n, steps, {"terrain.type": "plane", name: v})
                sweep[f"{name}={v:g}"] = {k: r[k] for k in ("fall_rate_first_episode", "falls_per_robot_minute", "mean_first_episode_length", "tracking_rmse", "mean_reward_per_step")}
                print(name, v, sweep[f"{name}={v:g}"]["fall_rate_first_episode"], flush=True)
        res["contact_parameter_sweep_plane"] = sweep
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = "own_actor_eval.json" if ACTOR_CHECKPOINT else "reference_actor_eval.json"
    with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
        json.dump(res, f, indent=1)
    print("written gpurun_out/" + name)


if __name__ == "__main__":
    main()
