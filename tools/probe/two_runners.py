import json, os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
from booster_gym_amd.utils.recorder import Recorder
def run(n, over, iters=8, warm=3):
    cfg = load_cfg("T1", dict({"env.num_envs": n, "basic.seed": 42}, **over)); cfg["runner"]["save_interval"] = 10 ** 9
    r = Runner(cfg=cfg); r.begin_training(Recorder(cfg, root=tempfile.mkdtemp(prefix="bg_loop_"), rank=0))
    for w in range(warm): r.train_iteration(w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(iters): r.train_iteration(warm + k)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / iters * 1e3
    r._flush_log()
    return r, ms
big = {"terrain.type": "trimesh", "sim.state_dtype": "fp16"}
mode = sys.argv[1]
def big_tensors(r):
    out = {}
    def visit(name, o, depth=0):
        if isinstance(o, torch.Tensor):
            if o.is_cuda and o.numel() * o.element_size() >= (1 << 20):
                out[name] = (o.data_ptr(), o.numel() * o.element_size())
        elif isinstance(o, (list, tuple)) and depth < 3:
            for i, v in enumerate(o): visit(f"{name}[{i}]", v, depth + 1)
        elif isinstance(o, dict) and depth < 3:
            for k, v in o.items(): visit(f"{name}.{k}", v, depth + 1)
    for owner, obj in (("runner", r), ("critic_tr", r._critic_tr), ("actor_tr", r._actor_tr), ("buffer", getattr(r.buffer, "tensors", None) or getattr(r.buffer, "_t", None) or {}),
                       ("wgrad", r._wgrad_group), ("opt", r.optimizer), ("env", r.env)):
        items = obj.items() if isinstance(obj, dict) else vars(obj).items()
        for k, v in items: visit(f"{owner}.{k}", v)
    return out
if mode == "ptrs":
    n = int(sys.argv[2]); prev = None
    for k in range(int(sys.argv[3])):
        r, ms = run(n, {"terrain.type": "plane"}, iters=6)
        cur = big_tensors(r)
        print(f"{n} envs, runner {k}: {ms:.2f} ms, {len(cur)} tensors >= 1 MB", flush=True)
        if prev is not None:
            for name in cur:
                if name in prev and prev[name][0] != cur[name][0]:
                    print(f"    moved: {name} {prev[name][1] >> 20} MB  {hex(prev[name][0])} -> {hex(cur[name][0])}  (mod 2 MiB: {prev[name][0] % (1 << 21)} -> {cur[name][0] % (1 << 21)})", flush=True)
        prev = cur
        del r
        torch.cuda.synchronize()
elif mode == "streams":  # advance torch's pool of high-priority streams before the first runner takes its side stream from it
    dummies = [torch.cuda.Stream(priority=-1) for _ in range(int(sys.argv[3]))]
    r, ms = run(int(sys.argv[2]), {"terrain.type": "plane"}, iters=6)
    print(f"{sys.argv[2]} envs behind {len(dummies)} unused high-priority streams: {ms:.2f} ms; side stream {r._side_stream}", flush=True)
elif mode == "repeat":  # the same configuration built again and again in one process: does where the allocator puts the buffers matter?
    n = int(sys.argv[2]); over = {"terrain.type": "plane"}
    for k in range(int(sys.argv[3])):
        r, ms = run(n, over, iters=10 if n <= 4096 else 6)
        ptrs = [r._critic_tr.acts[i].data_ptr() for i in range(3)] + [r._actor_tr.acts[i].data_ptr() for i in range(3)]
        print(f"{n} envs, runner {k}: {ms:.2f} ms   activation buffers at {[hex(p) for p in ptrs]}", flush=True)
        del r
        torch.cuda.synchronize()
elif mode == "alone":
    r2, ms = run(16384, big); print(f"16384 alone: {ms:.1f} ms", flush=True)
else:
    r1, ms1 = run(4096, {"terrain.type": "plane"}); print(f"4096 first: {ms1:.2f} ms", flush=True)
    r2, ms = run(16384, big); print(f"16384 with the 4096-env runner alive: {ms:.1f} ms", flush=True)
    del r2
    r3, ms = run(16384, big); print(f"16384 again: {ms:.1f} ms", flush=True)
print(torch.cuda.memory_allocated() / 2**30, "GiB allocated", torch.cuda.memory_reserved() / 2**30, "reserved")
