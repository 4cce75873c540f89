"""Diagnostic behind the tolerances of tests/parity_util.py: per field, the distribution over envs of the deviation of the HIP step from the
float64 oracle, next to the deviation of the oracle's own fp32 twin (libdynref32.so) on the same inputs.
python tools/parity_diag.py [plane|trimesh] [start_count] [n] -> gpurun_out/parity_diag_<terrain>.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_env as T
import parity_util as PU

terrain = sys.argv[1] if len(sys.argv) > 1 else "plane"
start = int(sys.argv[2]) if len(sys.argv) > 2 else 96
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cfg, env, ref = T._make(terrain, n)
ref32 = T._twin32(cfg, env, ref)
env.reset()
rng = np.random.default_rng(11)
for _ in range(15):
    env.step(torch.tensor(rng.uniform(-0.3, 0.3, (n, 12)), dtype=torch.float32, device=env.device))
env.common_step_counter = start
acc = {}
def add(name, side, v):
    acc.setdefault(name, {"gpu": [], "twin": []})[side].append(v)
for s in range(8):
    PU.sync_oracle(env, ref); PU.sync_oracle(env, ref32)
    act = rng.uniform(-0.6, 0.6, (n, 12)).astype(np.float32)
    obs, rew, done, extras = env.step(torch.tensor(act, device=env.device))
    o, p, r, d, t, terms, der = ref.step(act.astype(np.float64))
    o2, p2, r2, d2, t2, terms2, der2 = ref32.step(act.astype(np.float64))
    keep = (done.cpu().numpy() == d) & (d2 == d)
    fc = env.get_field("feet_contact").cpu().numpy() > 0.5
    same_g = (fc == np.asarray(der["feet_contact"]).astype(bool)).all(axis=1); same_t = (np.asarray(der2["feet_contact"]) == np.asarray(der["feet_contact"])).all(axis=1)
    G = {"root": (env.root_states.cpu().numpy(), ref32.root, ref.root), "dof_pos": (env.dof_pos.cpu().numpy(), ref32.q, ref.q),
         "dof_vel": (env.dof_vel.cpu().numpy(), ref32.qd, ref.qd), "torques": (env.get_field("torques").cpu().numpy(), der2["torques"], der["torques"]),
         "feet_pos": (env.get_field("feet_pos").cpu().numpy(), der2["feet_pos"].reshape(n, 6), der["feet_pos"].reshape(n, 6)),
         "obs": (obs.cpu().numpy(), o2, o), "priv": (extras["privileged_obs"].cpu().numpy(), p2, p)}
    for k, (g, w, f) in G.items():
        add(k, "gpu", PU.rel_state(g, f)[keep]); add(k, "twin", PU.rel_state(w, f)[keep])
    add("rew:reward", "gpu", PU.rel_reward(rew.cpu().numpy(), r)[keep]); add("rew:reward", "twin", PU.rel_reward(r2, r)[keep])
    for name in terms:
        kg, kt = (keep & same_g, keep & same_t) if name in PU.FLAG_TERMS else (keep, keep)
        add("rew:" + name, "gpu", PU.rel_reward(extras["rew_terms"][name].cpu().numpy(), terms[name])[kg]); add("rew:" + name, "twin", PU.rel_reward(terms2[name], terms[name])[kt])
    add("flagflip", "gpu", (~same_g[keep]).astype(float)); add("flagflip", "twin", (~same_t[keep]).astype(float))
out = {}
for k, v in acc.items():
    g, w = np.concatenate(v["gpu"]), np.concatenate(v["twin"])
    tol = PU.STATE_TOL.get(k, PU.REW_TOL)
    out[k] = {"n": int(g.size), "gpu": {"p50": float(np.median(g)), "p99": float(np.quantile(g, 0.99)), "p999": float(np.quantile(g, 0.999)), "max": float(g.max()), "frac_gt_tol": float((g > tol).mean())},
              "twin": {"p50": float(np.median(w)), "p99": float(np.quantile(w, 0.99)), "p999": float(np.quantile(w, 0.999)), "max": float(w.max()), "frac_gt_tol": float((w > tol).mean())}, "tol": tol}
    print(k.ljust(22), "tol %.0e | gpu p50 %.1e p99 %.1e p99.9 %.1e max %.1e >tol %.4f | twin p50 %.1e p99 %.1e p99.9 %.1e max %.1e >tol %.4f" % (
        tol, *[out[k]["gpu"][q] for q in ("p50", "p99", "p999", "max", "frac_gt_tol")], *[out[k]["twin"][q] for q in ("p50", "p99", "p999", "max", "frac_gt_tol")]), flush=True)
json.dump({"terrain": terrain, "num_envs": n, "steps": 8, "metric": "see tests/parity_util.py (rel_state / rel_reward)", "fields": out},
          open(os.path.join(ROOT, "gpurun_out", f"parity_diag_{terrain}.json"), "w"), indent=1)
