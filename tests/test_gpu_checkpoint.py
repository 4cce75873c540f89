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


def test_actor_kernels_and_export_reproduce_the_reference_deploy_codes_actions(tmp_path):
    """tests/golden/deploy_policy.npz: observations and network outputs of the reference's own deploy-side class (deploy/utils/policy.py:34-73) with its
    trained actor (weights as numbers: tests/golden/t1_actor.npz).  With those weights loaded, (1) the rollout's fused MFMA actor kernel (bg_actor_sample:
    the mean it samples around), (2) the update's chained forward + output layer and (3) the TorchScript that export_model.py writes -- the file that
    class loads on the robot -- give the reference's actions on the reference's observations."""
    import subprocess
    import sys

    from booster_gym_amd.utils.model import ActorCritic

    here = os.path.dirname(os.path.abspath(__file__))
    d = np.load(os.path.join(here, "golden", "deploy_policy.npz"))
    w = np.load(os.path.join(here, "golden", "t1_actor.npz"))
    obs = torch.tensor(d["obs"].reshape(-1, 47), device="cuda:0")
    raw = torch.tensor(d["raw_actions"].reshape(-1, 12), device="cuda:0")
    model = ActorCritic(12, 47, 14).to("cuda:0")
    model.actor.load_state_dict({k: torch.tensor(w[k]) for k in w.files})
    # (1) rollout inference kernel
    mu, act = torch.empty_like(raw), torch.empty_like(raw)
    model.sample_actions(obs, act, seed=1, counter=0, mu_out=mu)
    assert torch.allclose(mu, raw, rtol=0, atol=1e-5), (mu - raw).abs().max().item()
    # (2) the training-side forward (hand-written MFMA layers + library output layer)
    with torch.no_grad():
        assert torch.allclose(model.actor(obs), raw, rtol=0, atol=1e-5)
    # (3) checkpoint -> export_model.py -> TorchScript file, loaded the way deploy/utils/policy.py:9 loads it
    ck = tmp_path / "logs" / "run" / "nn"
    ck.mkdir(parents=True)
    torch.save({"model": {k: v.cpu() for k, v in model.state_dict().items()}}, str(ck / "model_1.pth"))
    root = os.path.dirname(here)
    p = subprocess.run([sys.executable, os.path.join(root, "export_model.py"), "--task=T1", "--checkpoint", str(ck / "model_1.pth")], cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    scripted = torch.jit.load(str(tmp_path / "deploy" / "models" / "T1.pt"))
    scripted.eval()
    with torch.no_grad():
        out = scripted(obs.cpu())
    assert torch.allclose(out, raw.cpu(), rtol=0, atol=1e-5), (out - raw.cpu()).abs().max().item()
    clip = torch.clamp(out, -1.0, 1.0)
    assert torch.allclose(clip, torch.tensor(d["actions"].reshape(-1, 12)), rtol=0, atol=1e-5)
