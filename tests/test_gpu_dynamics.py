"""K1 parity on the GPU: HIP forward dynamics (through the C ABI) vs the double-precision oracle.

Stated tolerance: relative 1e-4 on the accelerations for contact-free states with joints
inside or outside their limits; 5e-4 when stiff sole contacts are active (fp32 conditioning of k = 4e4 N/m).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _states(rng, m, n, contact):
    """contact: False = airborne, True = standing height (sole contacts), "low" = trunk 0.15-0.5 m above the ground in any orientation (the
    trunk box and the hip-yaw / shank cylinders touch as well), "crossed" = airborne with the hip rolls drawn inwards (leg against leg)."""
    crossed = contact in ("crossed", "crossed_gated")
    if crossed:
        contact = False
    root = np.zeros((n, 13))
    root[:, 2] = (rng.uniform(0.15, 0.5, n) if contact == "low" else rng.uniform(0.55, 0.72, n)) if contact else 5.0
    root[:, :2] = rng.uniform(-1, 1, (n, 2))
    ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, (3.0 if contact == "low" else 0.3) if contact else 1.0, n)
    root[:, 3:6] = ax * np.sin(ang / 2)[:, None]; root[:, 6] = np.cos(ang / 2)
    root[:, 7:13] = rng.normal(size=(n, 6)) * (0.3 if contact else 1.0)
    if contact:
        q = np.tile(np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2), (n, 1)) + rng.normal(size=(n, 12)) * 0.1
    else:
        q = rng.uniform(m.dof_lower - 0.05, m.dof_upper + 0.05, (n, 12))
    if crossed:
        q[:, 1], q[:, 7] = rng.uniform(-0.3, 0.0, n), rng.uniform(0.0, 0.3, n)
        q[:, [0, 6]], q[:, [3, 9]] = rng.uniform(-0.6, 0.2, (n, 2)), rng.uniform(0.0, 0.8, (n, 2))
    qd = rng.normal(size=(n, 12))
    tau = rng.uniform(-m.dof_effort, m.dof_effort, (n, 12))
    w = rng.normal(size=(n, 6)) * 10
    return root, q, qd, tau, w


@pytest.mark.parametrize("terrain,contact,tol,n", [("plane", False, 1e-4, 256), ("plane", True, 5e-4, 256), ("trimesh", True, 5e-4, 256), ("plane", "low", 1e-3, 256),
                                                   ("trimesh", "low", 1e-3, 256), ("plane", "crossed", 5e-4, 256),
                                                   # a ragged last block (100 = 3 x 32 + 4 envs) with every env's legs crossed: the item-parallel narrow
                                                   # phase runs five passes of seven envs per wave, the clamped lanes of the last wave vote with env 99
                                                   ("plane", "crossed", 5e-4, 100),
                                                   # ... and a few crossed envs among many (one pass, partly filled): crossed-leg states in every 9th env
                                                   ("plane", "sparse_crossed", 5e-4, 288),
                                                   # crossed legs with the trunk-low gate ON (terminate_height 0.05): the two-kernel launch, where kernel A
                                                   # defers every env whose legs can meet (SELF_DEFER), aba_compact_kernel lists them and kernel B evaluates
                                                   # the contacts lane per leg (SELF_INLINE) -- the path a config with body_gate_height > terminate_height ships
                                                   ("plane", "crossed_gated", 5e-4, 256)])
@pytest.mark.parametrize("packed", [False, True], ids=["leg_per_lane", "env_per_lane"])
def test_forward_dynamics_matches_oracle(flat_model, terrain, contact, tol, n, packed):
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg
    from oracle.dyn_ref import DynRef

    # rewards.terminate_height below the body-contact gate height switches the non-foot body contacts on (bg_env_cfg.body_gate_height): the ABA launch's
    # two-kernel scheme; the crossed-leg cases keep the shipped heights, where the launch is the ONE kernel with the narrow phase through LDS.
    # packed = bg_env_forward_dynamics_packed (one env per lane, both legs in 64-bit register pairs): shipped heights only (it has no body contacts)
    if packed and contact in ("low", "crossed_gated"):
        pytest.skip("the packed kernel carries no non-foot body contacts (refused with the trunk-low gate: test_packed_form_refuses_the_body_gate)")
    over = {} if contact in ("crossed", "sparse_crossed") or packed else {"rewards.terminate_height": 0.05}
    assert bool(over) == (contact in (False, True, "low", "crossed_gated") and not packed)
    cfg = load_cfg("T1", dict({"env.num_envs": n, "terrain.type": terrain}, **over))
    env = T1(cfg)
    tdict = None
    if terrain != "plane":
        t = env.terrain
        tdict = dict(height_field_raw=t.height_field_raw, hscale=t.horizontal_scale, vscale=t.vertical_scale, border_px=t.border_pixels)
    ref = DynRef(flat_model, feet_edge_pos=cfg["asset"]["feet_edge_pos"], terrain=tdict)
    twin = DynRef(flat_model, feet_edge_pos=cfg["asset"]["feet_edge_pos"], terrain=tdict, real="f32")  # the oracle's own source in single precision
    rng = np.random.default_rng(3)
    if contact == "sparse_crossed":
        root, q, qd, tau, w = _states(rng, flat_model, n, False)
        rc, qc, qdc, tc, wc = _states(rng, flat_model, n, "crossed")
        pick = np.arange(n) % 9 == 4
        root[pick], q[pick], qd[pick], tau[pick], w[pick] = rc[pick], qc[pick], qdc[pick], tc[pick], wc[pick]
    else:
        root, q, qd, tau, w = _states(rng, flat_model, n, contact)
    if terrain != "plane":
        root[:, 0] += 20.0; root[:, 1] += 5.0  # inside the rough strips
        root[:, 2] += np.array([ref.terrain_height(x, y) for x, y in root[:, :2]])
    dev = env.device
    f = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    qacc = env.forward_dynamics(f(root), f(q), f(qd), f(tau), f(w), packed=packed).cpu().numpy().astype(np.float64)
    cf = env.get_field("feet_contact_forces").cpu().numpy().reshape(n, 2, 3)
    worst, ncontact, nexcused, excused = 0.0, 0, 0, []
    for e in range(n):
        r32 = root[e].astype(np.float32).astype(np.float64)  # oracle sees the same rounded inputs
        args = (r32, q[e].astype(np.float32), qd[e].astype(np.float32), tau[e].astype(np.float32))
        kw = dict(base_wrench=w[e].astype(np.float32), mass_scale=env._mass_scale[e].astype(np.float32), com_off=env._com_off[e].astype(np.float32),
                  foot_mat=env._foot_mat[e].astype(np.float32).reshape(6))
        qa, cfr = ref.forward(*args, **kw)
        # any active contact is stiff (also a chance leg-against-leg one among the random airborne poses): 5e-4 at least
        raw = np.abs(qacc[e] - qa).max() / max(1.0, np.abs(qa).max())
        err = raw * (tol / max(tol, 5e-4) if np.abs(cfr).max() > 0 else 1.0)
        if err >= tol and np.abs(cfr).max() > 0:
            # a kN-range contact can make a state ill-conditioned in single precision altogether: then the oracle's OWN source built in float deviates
            # from its float64 build by more than the tolerance too, and the kernel is held to that deviation instead (x 1: no slack on top)
            qt, _ = twin.forward(*args, **kw)
            twin_err = np.abs(qt - qa).max() / max(1.0, np.abs(qa).max())
            assert raw <= twin_err, f"env {e}: relative qacc error {raw:.2e} beyond both the tolerance and the oracle's fp32 twin ({twin_err:.2e})"
            nexcused += 1
            excused.append(e)
            assert raw <= 3e-3, f"env {e}: relative qacc error {raw:.2e} above the absolute ceiling of an excused state"
            continue
        worst = max(worst, err)
        if np.abs(cfr).max() > 0:
            ncontact += 1
            assert np.abs(cf[e] - cfr[[6, 12]]).max() <= 2e-3 * max(1.0, np.abs(cfr).max())
    assert np.isfinite(qacc).all()
    assert worst < tol, f"worst relative qacc error {worst}"
    # the excused set is PINNED: one env of one case -- sparse_crossed: env 13, a 3.4 kN shank contact, twin 1.2e-3 -- and none anywhere else,
    # so a regression confined to stiff-contact states cannot hide behind the twin's own ill-conditioning
    assert excused == ([13] if contact == "sparse_crossed" and excused else []), (contact, nexcused, excused)
    if contact == "sparse_crossed":
        assert ncontact >= 8  # (32 of the 288 envs carry a crossed-leg state; 13 of them touch with this seed)
    elif contact:
        assert ncontact > n // 4


def test_packed_form_refuses_the_body_gate(flat_model):
    """bg_env_forward_dynamics_packed has no non-foot body contacts: an env created with the trunk-low gate (body_gate_height > terminate_height) is refused."""
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    env = T1(load_cfg("T1", {"env.num_envs": 64, "terrain.type": "plane", "rewards.terminate_height": 0.05}))
    z = lambda *s: torch.zeros(*s, device=env.device)
    root = z(64, 13); root[:, 2] = 0.7; root[:, 6] = 1.0
    with pytest.raises(RuntimeError, match="body contacts"):
        env.forward_dynamics(root, z(64, 12), z(64, 12), z(64, 12), packed=True)
    assert torch.isfinite(env.forward_dynamics(root, z(64, 12), z(64, 12), z(64, 12))).all()


@pytest.mark.parametrize("standing", [False, True], ids=["airborne", "standing_on_the_plane"])
@pytest.mark.parametrize("packed", [False, True], ids=["leg_per_lane", "env_per_lane"])
def test_mirrored_pose_gives_mirrored_accelerations_on_the_gpu(flat_model, tmp_path, standing, packed):
    """SURVEY section 8c, known-answer test (5) through the C ABI: on a SYMMETRISED copy of the model (conftest.symmetrised; loaded from a flat-model
    file, per-env randomisation of the inertials and foot materials off) a state that is its own mirror image gives mirrored joint accelerations,
    mirrored foot forces and no lateral / roll / yaw acceleration of the trunk.  In the lane-per-leg kernel the two legs are the two lanes of a pair
    and the trunk's share is split between them; in the packed kernel they are the halves of register pairs: a left / right slip in either shows at
    order one.  Tolerance: the parity tolerances of this file (fp32 sums in mirrored order differ in the last bits; stiff contacts amplify them)."""
    from conftest import MIRROR_SIGN, mirrored_states, symmetrised

    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    n = 256
    path = tmp_path / "T1_symmetrised.flat.json"
    symmetrised(flat_model).save(str(path))
    off = {f"randomization.{k}": None for k in ("base_com", "base_mass", "other_com", "other_mass", "friction", "compliance", "restitution")}
    env = T1(load_cfg("T1", dict({"env.num_envs": n, "terrain.type": "plane", "asset.file": str(path)}, **off)))
    assert np.all(env._mass_scale == 1.0) and np.all(env._com_off == 0.0) and np.ptp(env._foot_mat, axis=(0, 1)).max() == 0.0
    root, q, qd, tau = mirrored_states(np.random.default_rng(3), n, standing)
    f = lambda a: torch.tensor(a, dtype=torch.float32, device=env.device)
    qacc = env.forward_dynamics(f(root), f(q), f(qd), f(tau), f(np.zeros((n, 6))), packed=packed).cpu().numpy().astype(np.float64)
    cf = env.get_field("feet_contact_forces").cpu().numpy().reshape(n, 2, 3).astype(np.float64)
    S, my = np.array(MIRROR_SIGN), np.array([1.0, -1.0, 1.0])
    touching = np.abs(cf).max(axis=(1, 2)) > 0
    scale = np.maximum(1.0, np.abs(qacc).max(axis=1))
    joints = np.abs(qacc[:, 12:18] - S * qacc[:, 6:12]).max(axis=1) / scale
    trunk = np.abs(qacc[:, [1, 3, 5]]).max(axis=1) / scale
    feet = np.abs(cf[:, 1] - cf[:, 0] * my).max(axis=1) / np.maximum(1.0, np.abs(cf).max(axis=(1, 2)))
    print(f"mirror asymmetry on the GPU ({'packed' if packed else 'lane per leg'}): joints {joints.max():.2e}, trunk {trunk.max():.2e}, foot forces {feet.max():.2e}; "
          f"{int(touching.sum())} of {n} states with a contact")
    assert np.isfinite(qacc).all()
    tol = np.where(touching, 5e-4, 1e-4)
    assert (joints < tol).all() and (trunk < tol).all(), (joints.max(), trunk.max())
    assert (feet < 2e-3).all(), feet.max()
    assert touching.sum() >= (n // 2 if standing else n // 10)


def _quat_mul(a, b):  # xyzw, batched
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


@pytest.mark.parametrize("packed", [False, True], ids=["leg_per_lane", "env_per_lane"])
def test_size_independent_identities_at_4096_envs(flat_model, packed):
    """Properties of the forward dynamics that hold whatever the state, checked through the C ABI at the bench's size (4,096 envs, every per-env
    randomisation on, the shipped model) -- nothing here is compared with this build's oracle:
      * superposition: for a fixed state the accelerations are AFFINE in the joint torques, qacc(t1 + t2) - qacc(t1) - qacc(t2) + qacc(0) = 0
        (airborne states: joint-limit and leg-against-leg forces depend on the state only);
      * yaw / translation invariance on the plane: the whole state turned about the vertical by an angle and moved sideways gives the same joint
        accelerations, and trunk accelerations / foot forces turned by the same angle (standing states, sole contacts active)."""
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    n = 4096
    env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
    f = lambda a: torch.tensor(a, dtype=torch.float32, device=env.device)
    fd = lambda root, q, qd, tau, w: env.forward_dynamics(f(root), f(q), f(qd), f(tau), f(w), packed=packed).cpu().numpy().astype(np.float64)
    rng = np.random.default_rng(17)
    # --- superposition in the torques (airborne)
    root, q, qd, tau, w = _states(rng, flat_model, n, False)
    t2 = rng.uniform(-flat_model.dof_effort, flat_model.dof_effort, (n, 12))
    a12, a1, a2, a0 = fd(root, q, qd, tau + t2, w), fd(root, q, qd, tau, w), fd(root, q, qd, t2, w), fd(root, q, qd, np.zeros((n, 12)), w)
    scale = np.maximum(1.0, np.maximum(np.abs(a12), np.abs(a1) + np.abs(a2)).max(axis=1))
    sup = np.abs(a12 - a1 - a2 + a0).max(axis=1) / scale
    print(f"superposition residual at {n} envs ({'packed' if packed else 'lane per leg'}): max {sup.max():.2e}, 99.9 % {np.quantile(sup, 0.999):.2e}")
    assert np.isfinite(a12).all() and sup.max() < 5e-6, sup.max()  # (measured 4-5e-7)
    # --- yaw + translation invariance (standing on the plane)
    root, q, qd, tau, w = _states(rng, flat_model, n, True)
    psi, shift = rng.uniform(-np.pi, np.pi, n), rng.uniform(-3.0, 3.0, (n, 2))
    c, s = np.cos(psi), np.sin(psi)
    rotz = lambda v: np.stack([c * v[:, 0] - s * v[:, 1], s * v[:, 0] + c * v[:, 1], v[:, 2]], axis=1)
    r2 = root.copy()
    r2[:, :3] = rotz(root[:, :3]); r2[:, :2] += shift
    qz = np.stack([np.zeros(n), np.zeros(n), np.sin(psi / 2), np.cos(psi / 2)], axis=1)
    r2[:, 3:7] = _quat_mul(qz, root[:, 3:7])
    r2[:, 7:10], r2[:, 10:13] = rotz(root[:, 7:10]), rotz(root[:, 10:13])
    # (the base wrench is LOCAL_SPACE, envs/t1.py:522-527: it turns with the trunk by itself)
    qa = fd(root, q, qd, tau, w)
    cfa = env.get_field("feet_contact_forces").cpu().numpy().reshape(n, 2, 3).astype(np.float64)
    qb = fd(r2, q, qd, tau, w)
    cfb = env.get_field("feet_contact_forces").cpu().numpy().reshape(n, 2, 3).astype(np.float64)
    scale = np.maximum(1.0, np.abs(qa).max(axis=1))
    inv = np.maximum(np.abs(qb[:, 6:] - qa[:, 6:]).max(axis=1), np.maximum(np.abs(qb[:, :3] - rotz(qa[:, :3])).max(axis=1), np.abs(qb[:, 3:6] - rotz(qa[:, 3:6])).max(axis=1))) / scale
    fsc = np.maximum(1.0, np.abs(cfa).max(axis=(1, 2)))
    finv = np.maximum(np.abs(cfb[:, 0] - rotz(cfa[:, 0])).max(axis=1), np.abs(cfb[:, 1] - rotz(cfa[:, 1])).max(axis=1)) / fsc
    touching = np.abs(cfa).max(axis=(1, 2)) > 0
    print(f"yaw / translation invariance at {n} envs: accelerations max {inv.max():.2e} (99.9 % {np.quantile(inv, 0.999):.2e}), foot forces max {finv.max():.2e}; "
          f"{int(touching.sum())} envs with sole contact")
    assert touching.sum() > n // 4
    # (stiff contacts amplify the rounding of the turned inputs: the worst env sits an order of magnitude above the 99.9 % quantile)
    assert np.quantile(inv, 0.999) < 1e-4 and inv.max() < 1e-3, (np.quantile(inv, 0.999), inv.max())  # (measured 1e-5 / 1e-4)
    assert np.quantile(finv, 0.999) < 2e-4 and finv.max() < 2e-3, (np.quantile(finv, 0.999), finv.max())  # (measured 2e-5)
