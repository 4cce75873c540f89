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


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child(force, own=True):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BG_DIST_FORCE="1" if force else "0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", BG_OWN_RCCL="1" if own else "0")
    env.pop("BG_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[-1][len("RCCL_WORLD1 "):])


def test_rccl_collective_path_in_a_world_of_one():
    out = _child(force=True)
    assert out["sum_fp64_exact"] and out["avg_fp32_exact"] and out["dependent_ok"] and out["group_exact"], out
    assert out["max"] == 3.5 and out["broadcast_int"] == 1234 and abs(out["sync_grid"] - 0.3) < 1e-6, out
    assert out["iteration_finite"] and out["parameters_moved"] and out["shutdown"], out
    # the ranks' tail (sums, one grouped collective, optimiser launch with its own norm) against the single-process two-launch tail on the same seeded
    # iteration: the same update up to the order in which the squared norm is summed
    import numpy as np

    ref = _child(force=False)
    assert np.allclose(out["params"], ref["params"], rtol=0, atol=2e-6), np.abs(np.array(out["params"]) - np.array(ref["params"])).max()
    assert np.allclose(out["stats"], ref["stats"], rtol=1e-5, atol=1e-7) and abs(out["lr"] - ref["lr"]) < 1e-9, (out["stats"], ref["stats"], out["lr"], ref["lr"])


def test_process_groups_communicator_instead_of_the_own_one():
    """BG_OWN_RCCL=0 (the fallback should the own communicator misbehave on a node nobody here has seen, and what the first multi-GPU run is to be
    compared with): every per-mini-epoch exchange through torch.distributed's communicator -- the same exact collectives and the same iteration."""
    import numpy as np

    out, ref = _child(force=True, own=False), _child(force=True, own=True)
    assert out["own_rccl"] is False and ref["own_rccl"] is True
    assert out["sum_fp64_exact"] and out["avg_fp32_exact"] and out["dependent_ok"] and out["group_exact"] and out["iteration_finite"] and out["shutdown"], out
    assert np.allclose(out["params"], ref["params"], rtol=0, atol=1e-7), np.abs(np.array(out["params"]) - np.array(ref["params"])).max()
    assert np.allclose(out["stats"], ref["stats"], rtol=1e-6, atol=1e-9) and abs(out["lr"] - ref["lr"]) < 1e-12
