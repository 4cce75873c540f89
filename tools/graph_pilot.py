"""Do HIP graphs shrink the dead time between dependent launches of the training loop (verdict round 3, item 3)?

Pilot on the unchanged loop: Runner.rollout() (48 dependent launches) and Runner.update() (20 mini-epochs on two streams, ~25 launches each) are
captured AS THEY ARE into torch.cuda.CUDAGraph objects -- per-launch scalars (the action counter, the env's step count, Adam's step) stay the
values baked at capture time, which changes what the replays compute but not what they cost -- and the replays are timed against the eager
launches of the same code on the same box, HIP events on the launch stream.
    python tools/graph_pilot.py [num_envs=4096] -> gpurun_out/graph_pilot.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "basic.seed": 42})
r = Runner(cfg=cfg)
obs, infos = r.env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
for _ in range(3):
    r.iteration()
torch.cuda.synchronize()


def timed(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, host


out = {"num_envs": n}
for name, fn in (("rollout", r.rollout), ("update", r.update)):
    eager = [timed(fn) for _ in range(3)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            fn()  # warm-up on the capture stream
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = [timed(g.replay) for _ in range(3)]
        eager2 = [timed(fn) for _ in range(3)]
        out[name] = {"eager_gpu_ms": [round(a, 3) for a, _ in eager + eager2], "eager_host_ms": [round(b, 3) for _, b in eager + eager2],
                     "graph_gpu_ms": [round(a, 3) for a, _ in graph], "graph_host_ms": [round(b, 3) for _, b in graph]}
    except Exception as ex:  # capture not possible on this stack: say why
        out[name] = {"eager_gpu_ms": [round(a, 3) for a, _ in eager], "error": repr(ex)[:600]}
    print(name, json.dumps(out[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/graph_pilot.json", "w"), indent=1)
