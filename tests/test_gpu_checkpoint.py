"""checkpoint / export / play compatibility with the reference's formats (recorder.py:70-73, runner.py:82-97, 207-213,
export_model.py:26-29), exercised on the real runner."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner(n=64, epochs=2, **extra):
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    ov = {"env.num_envs": n, "terrain.type": "plane", "runner.mini_epochs": epochs}
    ov.update(extra)
    return Runner(cfg=load_cfg("T1", ov))


def test_checkpoint_roundtrip_and_reference_layout(tmp_path):
    from booster_gym_amd.utils.model import ActorCritic

    r = _runner()
    obs, infos = r.env.reset()
    r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
    r.iteration()
    ck = r.checkpoint_dict()
    assert set(ck.keys()) == {"model", "optimizer", "curriculum"} and tuple(ck["curriculum"].shape) == (21, 21)
    path = str(tmp_path / "model_1.pth")
    torch.save(ck, path)
    loaded = torch.load(path, map_location="cpu", weights_only=True)  # what the reference's _load does (runner.py:87)
    # the reference's own classes accept it: model keys, and a stock torch Adam takes the optimiser state
    m = ActorCritic(12, 47, 14)
    missing, unexpected = m.load_state_dict(loaded["model"], strict=False)
    assert not missing and not unexpected
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    opt.load_state_dict(loaded["optimizer"])
    st = opt.state_dict()["state"]
    assert len(st) == 17 and float(st[0]["step"]) == 2.0 and st[1]["exp_avg"].shape == m.critic[0].weight.shape
    # our runner resumes from it: same weights, same moments, same LR
    r2 = _runner(**{"basic.checkpoint": path})
    for (k, a), (_, b) in zip(r.model.state_dict().items(), r2.model.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    assert torch.equal(r.optimizer.exp_avg.cpu(), r2.optimizer.exp_avg.cpu()) and r2.optimizer.step_count == 2
    assert abs(r2.optimizer.lr.item() - r.optimizer.lr.item()) < 1e-12
    assert torch.equal(r.env.curriculum_prob.cpu(), r2.env.curriculum_prob.cpu())
    # learning rate after a resume, as the reference does it (runner.py:31-34,174-180): the restored lr serves the first optimiser step, then the
    # KL rule restarts from the yaml value (first mini-epoch KL = 0 -> x1.5) whatever the checkpoint held
    r3 = _runner(epochs=1, **{"basic.checkpoint": path})
    r3.optimizer.lr.fill_(3e-4)  # as if the checkpoint had been saved at a much larger adapted lr
    obs, infos = r3.env.reset()
    r3.buffer["obses"][0].copy_(obs); r3.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
    r3.iteration()
    assert abs(r3.optimizer.lr.item() - 1.5 * r3.cfg["algorithm"]["learning_rate"]) < 1e-9


def test_torchscript_export_matches_the_actor(tmp_path):
    """export_model.py:26-29: torch.jit.script(model.actor) -- what deploy/utils/policy.py:9 loads."""
    r = _runner()
    scripted = torch.jit.script(r.model.actor.cpu())
    p = str(tmp_path / "T1.pt")
    scripted.save(p)
    loaded = torch.jit.load(p)
    x = torch.randn(5, 47)
    assert torch.allclose(loaded(x), r.model.actor(x), atol=1e-6)
    assert [k for k, _ in loaded.state_dict().items()] == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "6.weight", "6.bias"]


def test_play_loop_runs_deterministic_actions(tmp_path):
    r = _runner()
    rec = str(tmp_path / "traj.npz")
    steps = r.play(max_steps=25, record_path=rec)
    d = np.load(rec)
    assert steps == 25 and d["root"].shape == (25, 13) and np.isfinite(d["root"]).all()


def test_train_entry_writes_the_reference_log_layout(tmp_path, monkeypatch):
    """Runner.train(): logs/<ts>/{config.yaml, nn/model_<it>.pth, summaries/} with the reference's scalar names (runner.py:190-204)."""
    monkeypatch.chdir(tmp_path)
    r = _runner(**{"basic.max_iterations": 2, "runner.save_interval": 2})
    r.train()
    runs = os.listdir(tmp_path / "logs")
    assert len(runs) == 1
    base = tmp_path / "logs" / runs[0]
    assert (base / "config.yaml").is_file() and (base / "nn" / "model_2.pth").is_file()
    tags = {__import__("json").loads(l)["tag"] for l in open(base / "summaries" / "scalars.jsonl")}
    for t in ("value_loss", "actor_loss", "bound_loss", "entropy", "kl_mean", "lr", "curriculum/mean_lin_vel_level", "curriculum/max_ang_vel_level", "steps",
              "reward", "episode/survival", "episode/feet_swing"):
        assert t in tags, t
    assert "episode/feet_vel_z" not in tags  # dropped reward terms are not logged
