"""`python bench.py --gpus N` must produce the N-rank bench line on its own (the driver's multi-GPU command): the process becomes a launcher
that starts the ranks as children before anything touches the GPU.  On the one-GPU test box the two ranks share device 0 and talk over gloo
(BG_LOCAL_DEVICE / BG_DIST_BACKEND); on a multi-GPU node the same command runs one rank per GPU over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command_line(monkeypatch):
    """Host logic, no GPU: the launcher re-invokes this file under torch.distributed.run with the caller's flags and a loopback rendezvous."""
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "2"])
    cmd = seen["cmd"]
    assert rc == 7  # the children's status is the launcher's
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_gpus_2_prints_one_line():
    env = dict(os.environ, BG_DIST_BACKEND="gloo", BG_LOCAL_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--num_envs", "256",
                        "--no-cpu-baseline", "--no-extra"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 2 * 256 * 24 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["phase_ms"]["all_reduce_ms"] > 0
