"""RCCL (torch.distributed backend "nccl") executed on the one-GPU box: a world of one rank takes the whole collective path of
booster_gym_amd/utils/parallel.py and of Runner.update() (the collective sits where the reference steps its optimiser,
utils/runner.py:162-165).  What this does NOT cover: more than one rank on RCCL (xGMI transport, rendezvous of several processes) -- that is the
driver's multi-GPU run; the arithmetic of several ranks is covered under gloo (tests/test_host_logic.py, tests/test_gpu_dp.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_collective_path_in_a_world_of_one():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", BG_DIST_FORCE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BG_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")]
    assert line, p.stdout[-2000:]
    out = json.loads(line[-1][len("RCCL_WORLD1 "):])
    assert out["sum_fp64_exact"] and out["avg_fp32_exact"] and out["dependent_ok"], out
    assert out["max"] == 3.5 and out["broadcast_int"] == 1234 and abs(out["sync_grid"] - 0.3) < 1e-6, out
    assert out["iteration_finite"] and out["parameters_moved"] and out["shutdown"], out
