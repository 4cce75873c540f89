"""Data-parallel PPO on real kernels: two ranks (gloo backend, both on the one GPU of the test box) each simulate 64 envs and run
Runner.update(); the result must equal the reference update loop (oracle/ppo_ref.py) on the UNION of both ranks' rollouts from the same
initial weights, and both ranks must end with identical parameters.  (On a multi-GPU node the same code runs over RCCL.)"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException as ex:  # the parent fails fast with the real error instead of a queue time-out
        import traceback

        q.put(("error", rank, "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BG_DIST_BACKEND="gloo", BG_LOCAL_DEVICE="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.runner import Runner
    from oracle.ppo_ref import ppo_update_reference

    E, n = 2, 64
    cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "runner.mini_epochs": E})
    r = Runner(cfg=cfg)
    assert r.world_size == 2 and r.rank == rank
    obs, infos = r.env.reset()
    r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
    r.rollout()
    T = cfg["runner"]["horizon_length"]
    sd0 = {k: v.detach().clone() for k, v in r.model.state_dict().items()}
    b = r.buffer

    def gather(t, dim):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        return torch.cat(parts, dim=dim)

    full = {k: gather(b[k].to(torch.uint8) if b[k].dtype == torch.bool else b[k], 1) for k in ("obses", "privileged_obses", "actions", "rewards", "dones", "time_outs")}
    acc = r.update()
    summ = r._summarize(acc)
    flat = torch.cat([p.detach().reshape(-1) for p in r.model.parameters()])
    other = gather(flat.view(1, -1), 0)
    ref_model = ActorCritic(12, 47, 14).to(r.device)
    ref_model.load_state_dict(sd0)
    stats_ref, lr_ref = ppo_update_reference(ref_model, torch.optim.Adam(ref_model.parameters(), lr=1e-5), full["obses"][:T], full["privileged_obses"][:T],
                                             full["actions"], full["rewards"].clone(), full["dones"].bool(), full["time_outs"].bool(), full["obses"][T],
                                             full["privileged_obses"][T], mini_epochs=E, learning_rate=1e-5)
    ref_flat = torch.cat([p.detach().reshape(-1) for p in ref_model.parameters()])
    q.put((rank, float((other[0] - other[1]).abs().max()), float((flat - ref_flat).abs().max()), float((flat - torch.cat([v.reshape(-1) for v in [sd0[k] for k, _ in r.model.named_parameters()]])).abs().max()),
           summ["kl_mean"], stats_ref["kl_mean"], summ["value_loss"], stats_ref["value_loss"], summ["lr"], lr_ref, bool((obs != 0).any())))
    r.dp.shutdown()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_update_equals_reference_on_the_union():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = []
        for _ in procs:
            item = q.get(timeout=300)
            if item[0] == "error":
                pytest.fail(f"rank {item[1]} raised:\n{item[2]}")
            res.append(item)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:  # never leave a rank behind holding the GPU, the port and a blocked collective
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    for rank, rank_diff, ref_diff, moved, kl, kl_ref, vl, vl_ref, lr, lr_ref, ok in res:
        assert ok
        assert rank_diff == 0.0, "ranks diverged"
        assert moved > 1e-6, "parameters did not change"
        # Adam normalises every gradient component to a step of ~lr: a parameter whose gradient is at rounding level can step either way, so
        # after 20 steps of 1e-5 the bound is a fraction of one step, not a relative error of the gradients
        assert ref_diff < 5e-6 + 1e-3 * moved, (rank, ref_diff, moved)
        assert abs(kl - kl_ref) <= 2e-4 * max(1.0, abs(kl_ref)) and abs(vl - vl_ref) <= 2e-4 * max(1.0, abs(vl_ref))
        assert abs(lr - lr_ref) < 1e-9
