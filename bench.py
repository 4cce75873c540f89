"""Headline benchmark: env-steps/sec of the full PPO training loop (rollout + update), T1, 4096 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one PPO iteration on synthetic (randomly initialised) policy weights: 24 env-steps x 4096 envs per GPU through the
HIP simulator + 20 full-batch optimiser steps (BASELINE.json configs[1], flat terrain).  Rank 0 prints
ONE JSON line.  `value` = world * N * T * K / wall (max over ranks).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak (same guide)

# Algorithmic HBM bytes of the fused env-step kernel per env per env-step: every per-env field it reads
# or writes once per launch, 4 bytes each.
ENV_STEP_BYTES = 4 * (
    # reads: root 13, q/qd 24, last targets/actions/qd 36, last root vel 6, cmd/gait 5, filters 6, last feet 6, push 6,
    #        kp/kd/friction 36, mass scale 13, com offset 39, foot material 6, base_mass_scaled 4, origin 3, int state 4, actions 12
    (13 + 24 + 36 + 6 + 5 + 6 + 6 + 6 + 36 + 13 + 39 + 6 + 4 + 3 + 4 + 12)
    # writes: root 13, q/qd 24, last targets 12, actions/last actions 24, last qd 12, last root vel 6, cmd/gait 5, filters 6,
    #         last feet 6, push 6, contact 6, derived (feet pos/roll/yaw/contact 12, torques 12, base vel/gravity 9), episode sums 27,
    #         int state 4, obs 47, privileged 14, reward terms 26, rew 1
    + (13 + 24 + 12 + 24 + 12 + 6 + 5 + 6 + 6 + 6 + 6 + 33 + 27 + 4 + 47 + 14 + 26 + 1)
) + 2  # done + time_out bytes


def gemm_flops_per_iteration(n_envs, horizon, mini_epochs):
    """GEMM flops of one update phase: fwd + bwd (= 3 x fwd) of actor and critic over the full batch per
    mini-epoch, plus the no-grad passes (old_mu once, last values every mini-epoch)."""
    actor = 2 * (47 * 256 + 256 * 128 + 128 * 128 + 128 * 12)
    critic = 2 * (61 * 256 + 256 * 256 + 256 * 128 + 128 * 1)
    B = n_envs * horizon
    train = mini_epochs * 3 * B * (actor + critic)
    nograd = B * actor + mini_epochs * n_envs * critic  # old_mu once, last_values every mini-epoch
    return train + nograd


def cpu_baseline(cfg, n_sample=128, horizon=24, mini_epochs=20):
    """CPU restatement baseline ("port"): oracle physics (C, OpenMP over envs) for a 24-step rollout of `n_sample` envs with
    the actor evaluated in torch-CPU, plus the 20 mini-epoch PPO update in torch-CPU on that batch.  NOT PhysX."""
    import numpy as np

    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.urdf import load_model
    from oracle.dyn_ref import DynRef
    from oracle.ppo_ref import ppo_update_reference

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    m = load_model(cfg["asset"]["file"])
    ref = DynRef(m, feet_edge_pos=cfg["asset"]["feet_edge_pos"])
    rng = np.random.default_rng(0)
    n = n_sample
    model = ActorCritic(12, 47, 14)
    root = np.zeros((n, 13)); root[:, 2] = 0.72; root[:, 6] = 1.0
    default = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2)
    q = np.tile(default, (n, 1)); qd = np.zeros((n, 12)); last_t = q.copy()
    kp = np.tile(np.array([200, 200, 200, 200, 50, 50] * 2, dtype=float), (n, 1)); kd = np.tile(np.array([5, 5, 5, 5, 1, 1] * 2, dtype=float), (n, 1))
    fric = np.zeros((n, 12)); ms = np.ones((n, 13)); co = np.zeros((n, 39)); fm = np.tile(np.array([1.0, 1.0, 0.0] * 2), (n, 1))
    delay = np.zeros(n, dtype=np.int32); wrench = np.zeros((n, 6))
    obs = torch.zeros(horizon, n, 47); priv = torch.zeros(horizon, n, 14); acts = torch.zeros(horizon, n, 12)
    t0 = time.perf_counter()
    for t in range(horizon):
        o = torch.zeros(n, 47)
        o[:, 11:23] = torch.tensor(q - default, dtype=torch.float32); o[:, 23:35] = torch.tensor(qd * 0.1, dtype=torch.float32)
        with torch.no_grad():
            a = torch.distributions.Normal(model.actor(o), torch.exp(model.logstd)).sample().clamp(-1, 1)
        obs[t], acts[t] = o, a
        ref.substeps_batch(10, ms, co, fm, kp, kd, fric, m.dof_effort, root, q, qd, default + a.numpy().astype(np.float64), last_t, delay, wrench)
    t_roll = time.perf_counter() - t0
    rew = torch.rand(horizon, n); dones = torch.zeros(horizon, n, dtype=torch.bool); touts = torch.zeros(horizon, n, dtype=torch.bool)
    t0 = time.perf_counter()
    ppo_update_reference(model, torch.optim.Adam(model.parameters(), lr=1e-5), obs, priv, acts, rew, dones, touts, obs[-1], priv[-1],
                         mini_epochs=mini_epochs)
    t_upd = time.perf_counter() - t0
    total = t_roll + t_upd
    return {"value": n * horizon / total, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {horizon} env-steps: oracle/dyn_ref.c physics (OpenMP) + torch-CPU actor, then {mini_epochs} torch-CPU PPO "
                      f"mini-epochs on that batch; rollout {t_roll:.2f}s update {t_upd:.2f}s; obs/reward task logic not included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--terrain", type=str, default="plane")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist

    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    cfg = load_cfg("T1", {"env.num_envs": args.num_envs, "terrain.type": args.terrain, "basic.seed": 42})
    runner = Runner(cfg=cfg)  # initialises the process group when WORLD_SIZE > 1
    T, N, E = cfg["runner"]["horizon_length"], runner.env.num_envs, cfg["runner"]["mini_epochs"]
    dev = runner.device

    obs, infos = runner.env.reset()
    runner.buffer["obses"][0].copy_(obs)
    runner.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.iteration()
    # instrument: HIP events around every env-step launch and around the update phase (torch's current stream is the launch stream)
    step_events, phase_events = [], []
    orig_step_to = runner.env.step_to

    def timed_step_to(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig_step_to(*a, **k); e1.record()
        step_events.append((e0, e1))

    runner.env.step_to = timed_step_to
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r0, r1, r2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        r0.record(); runner.rollout(); r1.record(); runner.update(); r2.record()
        runner.buffer["obses"][0].copy_(runner.buffer["obses"][T]); runner.buffer["privileged_obses"][0].copy_(runner.buffer["privileged_obses"][T])
        phase_events.append((r0, r1, r2))
    barrier()
    wall = time.perf_counter() - t0
    tw = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())

    if rank == 0:
        step_ms = sum(a.elapsed_time(b) for a, b in step_events) / max(len(step_events), 1)
        roll_ms = sum(a.elapsed_time(b) for a, b, _ in phase_events) / len(phase_events)
        upd_ms = sum(b.elapsed_time(c) for _, b, c in phase_events) / len(phase_events)
        env_bytes = N * ENV_STEP_BYTES
        sim_gbs = env_bytes / (step_ms * 1e-3) / 1e9
        flops = gemm_flops_per_iteration(N, T, E)
        stats = runner.env.episode_stats(reset=False).cpu().tolist()
        out = {
            "metric": "env-steps/sec (whole node), PPO rollout+update, T1 4096 envs/GPU",
            "value": world * N * T * args.steps / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (random-init policy, seeded domain randomisation)",
            "config": {"workload": f"T1 {args.terrain} terrain, {N} envs/GPU, horizon {T}, {E} mini-epochs, full batch (BASELINE.json configs[1])",
                       "envs_per_gpu": N, "parallelism": f"dp{world}"},
            "ppo_iters_per_s": args.steps / wall,
            "phase_ms": {"rollout": roll_ms, "update": upd_ms},
            "roofline": {"kernel": "env_step_kernel (fused 10 substeps + task logic)", "bound": "hbm", "achieved": sim_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": sim_gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": step_ms * 1e3,
                         "algorithmic_bytes_per_launch": env_bytes},
            "roofline_update": {"bound": "mfma", "achieved": flops / (upd_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "note": "actor+critic GEMM flops of the update phase / update-phase time (which also holds GAE, loss, Adam)"},
            "nonfinite_resets": stats[-1],
        }
        out["roofline_update"]["frac"] = out["roofline_update"]["achieved"] / MFMA_F32_PEAK_TF
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as ex:  # the bench line must still print
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
