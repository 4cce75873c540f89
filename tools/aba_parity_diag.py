"""Per-env deviation of the ABA launch from the float64 oracle on the states of tests/test_gpu_dynamics.py (diagnostic: which envs carry the
worst relative error, with or without contact).   python tools/aba_parity_diag.py [case] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from booster_gym_amd.envs import T1
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.urdf import FlatModel
from oracle.dyn_ref import DynRef
from test_gpu_dynamics import _states
case = sys.argv[1] if len(sys.argv) > 1 else "sparse_crossed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 288
m = FlatModel.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'booster_gym_amd', 'resources', 'T1', 'T1_locomotion.flat.json'))
cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"})
env = T1(cfg)
ref = DynRef(m, feet_edge_pos=cfg["asset"]["feet_edge_pos"], terrain=None)
twin = DynRef(m, feet_edge_pos=cfg["asset"]["feet_edge_pos"], terrain=None, real="f32")  # the oracle's own source in single precision
rng = np.random.default_rng(3)
if case == "sparse_crossed":
    root, q, qd, tau, w = _states(rng, m, n, False)
    rc, qc, qdc, tc, wc = _states(rng, m, n, "crossed")
    pick = np.arange(n) % 9 == 4
    root[pick], q[pick], qd[pick], tau[pick], w[pick] = rc[pick], qc[pick], qdc[pick], tc[pick], wc[pick]
else:
    root, q, qd, tau, w = _states(rng, m, n, {"airborne": False, "standing": True}.get(case, case))
f = lambda a: torch.tensor(a, dtype=torch.float32, device=env.device)
packed = os.environ.get("BG_ABA_PK", "1") != "0"
qacc = env.forward_dynamics(f(root), f(q), f(qd), f(tau), f(w), packed=packed).cpu().numpy().astype(np.float64)
rows = []
for e in range(n):
    qa, cfr = ref.forward(root[e].astype(np.float32).astype(np.float64), q[e].astype(np.float32), qd[e].astype(np.float32), tau[e].astype(np.float32),
                          base_wrench=w[e].astype(np.float32), mass_scale=env._mass_scale[e].astype(np.float32), com_off=env._com_off[e].astype(np.float32),
                          foot_mat=env._foot_mat[e].astype(np.float32).reshape(6))
    kw = dict(base_wrench=w[e].astype(np.float32), mass_scale=env._mass_scale[e].astype(np.float32), com_off=env._com_off[e].astype(np.float32),
              foot_mat=env._foot_mat[e].astype(np.float32).reshape(6))
    qt, _ = twin.forward(root[e].astype(np.float32).astype(np.float64), q[e].astype(np.float32), qd[e].astype(np.float32), tau[e].astype(np.float32), **kw)
    rows.append((np.abs(qacc[e] - qa).max() / max(1.0, np.abs(qa).max()), e, float(np.abs(cfr).max()), float(np.abs(qa).max()), int(np.abs(qacc[e] - qa).argmax()),
                 np.abs(qt - qa).max() / max(1.0, np.abs(qa).max())))
rows.sort(reverse=True)
print("packed (env per lane)" if packed else "leg per lane", case, n)
for r in rows[:8]:
    print(f"  env {r[1]:4d} rel err {r[0]:.2e} (fp32 twin of the oracle: {r[5]:.2e}) contact {r[2]:9.2f} N  max|qacc| {r[3]:9.1f}  worst component {r[4]}")
print("  median rel err %.2e" % np.median([r[0] for r in rows]))
