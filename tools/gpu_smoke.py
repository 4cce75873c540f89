"""Quick GPU exercise: trained reference actor walking in the HIP simulator + step timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1

def actor_from_npz(dev):
    W = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "t1_actor.npz"))
    layers = []
    for i in (0, 2, 4, 6):
        w = torch.tensor(W[f"{i}.weight"], device=dev); b = torch.tensor(W[f"{i}.bias"], device=dev)
        layers.append((w, b))
    def f(o):
        x = o
        for k, (w, b) in enumerate(layers):
            x = x @ w.T + b
            if k < 3: x = torch.nn.functional.elu(x)
        return x
    return f

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    terrain = sys.argv[2] if len(sys.argv) > 2 else "plane"
    cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": terrain})
    env = T1(cfg)
    pi = actor_from_npz(env.device)
    obs, extras = env.reset()
    torch.cuda.synchronize()
    print("reset ok; obs finite:", bool(torch.isfinite(obs).all()), "obs[0,:12]", obs[0, :12].cpu().numpy().round(3))
    tot_done = 0
    t0 = time.time()
    for s in range(300):
        act = pi(obs)
        obs, rew, done, extras = env.step(act)
        tot_done += int(done.sum())
        if s % 50 == 0:
            root = env.root_states
            print(f"step {s}: rew {rew.mean().item():.4f} done {int(done.sum())} z {root[:,2].mean().item():.3f} "
                  f"vx_body {env.get_field('base_lin_vel')[:,0].mean().item():.3f} cmdx {env.commands[:,0].mean().item():.3f} finite {bool(torch.isfinite(obs).all())}")
    torch.cuda.synchronize()
    print("300 steps wall", time.time() - t0, "total dones", tot_done, "of", n)
    st = env.episode_stats(reset=False).cpu().numpy()
    print("episode stats: finished", st[0], "mean len", st[1] / max(st[0], 1), "mean rew", st[2] / max(st[0], 1))
    # timing of the bare env step
    act = torch.zeros(n, 12, device=env.device)
    for _ in range(5): env.step(act)
    torch.cuda.synchronize(); t0 = time.time()
    K = 50
    for _ in range(K): env.step(act)
    torch.cuda.synchronize(); dt = (time.time() - t0) / K
    print(f"env.step: {dt*1e6:.1f} us per step at N={n} -> {n/dt/1e6:.2f} M env-steps/s (sim only)")

if __name__ == "__main__":
    main()
