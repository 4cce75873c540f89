"""oracle/cpu_baseline.py -- TEST INFRASTRUCTURE: times the CPU restatement for bench.py's `cpu_baseline` leg.

Run as a child process (`python -m oracle.cpu_baseline`) so that the OpenMP runtime of the C oracle and torch's CPU thread
pool live in a process that never touched the GPU.  Prints one JSON object.
Workload = a bounded sample of bench.py's step: `n` envs x 24 env-steps (oracle/dyn_ref.c physics with OpenMP over envs, actor in
torch-CPU) followed by 20 PPO mini-epochs in torch-CPU (oracle/ppo_ref.py) on that batch.  This is NOT PhysX-CPU.
"""
import json
import os
import sys
import time


def _cpu_share():
    """Threads this process may really use: the affinity mask, capped at 16 (the GPU box gives 16 cores per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


_THREADS = int(os.environ.get("BG_CPU_THREADS", _cpu_share()))
os.environ["OMP_NUM_THREADS"] = str(_THREADS)  # before numpy / torch / the oracle's libgomp start their pools
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(n=256, horizon=24, mini_epochs=20):
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.urdf import load_model
    from oracle.dyn_ref import DynRef
    from oracle.ppo_ref import ppo_update_reference

    cfg = load_cfg("T1")
    cores = _THREADS
    torch.set_num_threads(cores)
    m = load_model(cfg["asset"]["file"])
    ref = DynRef(m, feet_edge_pos=cfg["asset"]["feet_edge_pos"])
    torch.manual_seed(0)
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
    print(json.dumps({"value": n * horizon / total, "unit": "env-steps/s", "cores": cores, "kind": "port",
                      "sample": f"{n} envs x {horizon} env-steps: oracle/dyn_ref.c physics (C, OpenMP over envs) + torch-CPU actor, then "
                                f"{mini_epochs} torch-CPU PPO mini-epochs on that batch; rollout {t_roll:.2f}s, update {t_upd:.2f}s; "
                                "obs/reward task logic not included; CPU restatement baseline (not PhysX)"}))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
