"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE: times the CPU restatement for bench.py's `cpu_baseline` leg.

Run as a child process (`python -m oracle.cpu_baseline <num_envs>`) so that the OpenMP runtime of the C oracle and torch's CPU thread pool
live in a process that never touched the GPU.  Prints one JSON object.

Workload = bench.py's step on the host: `num_envs` envs x 24 env-steps of the FULL env step -- physics (oracle/dyn_ref.c in its single-precision
build libdynref32.so, OpenMP over envs: BASELINE.md section 3 plans the CPU baseline in fp32 like the reference's PhysX-CPU path) + the task
logic of oracle/task_ref.py (observations, 23 reward terms, termination, resets, command resampling, noise; numpy) + the torch-CPU actor --
followed by 20 full-batch PPO mini-epochs in torch-CPU (oracle/ppo_ref.py) on that batch.  This is the build's own restatement, NOT PhysX-CPU
(Isaac Gym is absent from the image), and it is a reported baseline, not a target.
"""
import json
import os
import platform
import sys
import time


def _cpu_share():
    """Threads this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one (cgroup v2 cpu.max, v1
    cpu.cfs_quota_us / cpu.cfs_period_us).  Returns (threads, where the number comes from)."""
    try:
        n, src = len(os.sched_getaffinity(0)), "affinity mask"
    except AttributeError:
        n, src = os.cpu_count() or 1, "os.cpu_count"
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, src = max(1, int(quota + 0.5)), f"cgroup CPU quota ({quota:.1f} cores) below the affinity mask"
    return max(1, n), src


_SHARE, _SHARE_SOURCE = _cpu_share()
_THREADS = int(os.environ.get("BG_CPU_THREADS", _SHARE))
os.environ["OMP_NUM_THREADS"] = str(_THREADS)  # before numpy / torch / the oracle's libgomp start their pools
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main(n=4096, horizon=24, mini_epochs=20):
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.urdf import load_model
    from oracle.dyn_ref import DynRef
    from oracle.ppo_ref import ppo_update_reference
    from oracle.task_ref import T1Ref

    cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "basic.seed": 42})  # BASELINE.json configs[1]: flat terrain
    cores = _THREADS
    torch.set_num_threads(cores)
    m = load_model(cfg["asset"]["file"])
    dyn = DynRef(m, feet_edge_pos=cfg["asset"]["feet_edge_pos"], real="f32",
                 phys={"terrain_mu": 0.5 * (cfg["terrain"]["static_friction"] + cfg["terrain"]["dynamic_friction"]), "terrain_restitution": cfg["terrain"]["restitution"]})
    # nominal per-env parameters (PD gains by the substring rule of t1.py:72-80; randomisation draws do not change the cost)
    kp, kd = np.zeros(12), np.zeros(12)
    for i, name in enumerate(m.dof_names):
        for key, v in cfg["control"]["stiffness"].items():
            if key in name:
                kp[i] = v
        for key, v in cfg["control"]["damping"].items():
            if key in name:
                kd[i] = v
    rep = lambda a: np.tile(np.asarray(a, dtype=np.float64), (n, 1))
    side = int(np.ceil(np.sqrt(n)))
    origins = np.stack([(np.arange(n) // side) * 1.0, (np.arange(n) % side) * 1.0, np.zeros(n)], axis=1)
    params = dict(kp=rep(kp), kd=rep(kd), fric=np.zeros((n, 12)), mass_scale=np.ones((n, 13)), com_off=np.zeros((n, 39)),
                  foot_mat=rep([1.0, 1.0, 0.0] * 2), bms=np.zeros((n, 4)), origins=origins)
    env = T1Ref(cfg, m, dyn, params, terrain=None, seed=42)
    torch.manual_seed(0)
    model = ActorCritic(12, 47, 14)
    obs = torch.zeros(horizon + 1, n, 47); priv = torch.zeros(horizon + 1, n, 14); acts = torch.zeros(horizon, n, 12)
    rew = torch.zeros(horizon, n); dones = torch.zeros(horizon, n, dtype=torch.bool); touts = torch.zeros(horizon, n, dtype=torch.bool)
    o, p = env.reset()
    obs[0], priv[0] = torch.tensor(o, dtype=torch.float32), torch.tensor(p, dtype=torch.float32)
    t_phys = 0.0
    sub = dyn.substeps_batch

    def timed_sub(*a, **k):
        nonlocal t_phys
        t = time.perf_counter()
        r = sub(*a, **k)
        t_phys += time.perf_counter() - t
        return r

    dyn.substeps_batch = timed_sub
    t0 = time.perf_counter()
    for t in range(horizon):  # runner.py:106-121
        with torch.no_grad():
            a = torch.distributions.Normal(model.actor(obs[t]), torch.exp(model.logstd)).sample()
        acts[t] = a
        o, p, r, d, to, _, _ = env.step(a.numpy().astype(np.float64))
        obs[t + 1], priv[t + 1] = torch.tensor(o, dtype=torch.float32), torch.tensor(p, dtype=torch.float32)
        rew[t], dones[t], touts[t] = torch.tensor(r, dtype=torch.float32), torch.tensor(d), torch.tensor(to)
    t_roll = time.perf_counter() - t0
    t0 = time.perf_counter()
    ppo_update_reference(model, torch.optim.Adam(model.parameters(), lr=1e-5), obs[:horizon], priv[:horizon], acts, rew, dones, touts, obs[horizon],
                         priv[horizon], mini_epochs=mini_epochs)
    t_upd = time.perf_counter() - t0
    total = t_roll + t_upd
    print(json.dumps({"value": n * horizon / total, "unit": "env-steps/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
                      "host_cores_visible": os.cpu_count(), "cores_source": "BG_CPU_THREADS" if "BG_CPU_THREADS" in os.environ else _SHARE_SOURCE, "num_envs": n,
                      "phase_s": {"rollout": t_roll, "rollout_physics_only": t_phys, "update": t_upd},
                      "sample": f"{n} envs x {horizon} env-steps, task logic included (observations, rewards, termination, resets, resampling, noise: "
                                f"oracle/task_ref.py) around fp32 physics (oracle/dyn_ref.c as libdynref32.so, OpenMP over envs) + torch-CPU actor, then "
                                f"{mini_epochs} full-batch torch-CPU PPO mini-epochs on that batch; rollout {t_roll:.2f}s (physics {t_phys:.2f}s), update {t_upd:.2f}s; "
                                f"{cores} threads on {_cpu_model()}; CPU restatement baseline (not PhysX)"}))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
