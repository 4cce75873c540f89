"""Edge cases and size-independent properties of the HIP env (through the C ABI): ragged / tiny / huge env counts, determinism,
independence of environments, state invariants, the non-finite guard, dropped reward terms, argument errors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NO_DR = {"randomization.dof_stiffness": None, "randomization.dof_damping": None, "randomization.dof_friction": None, "randomization.friction": None,
         "randomization.compliance": None, "randomization.restitution": None, "randomization.base_com": None, "randomization.base_mass": None,
         "randomization.other_com": None, "randomization.other_mass": None}


def _env(n, terrain="plane", extra=None):
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    ov = {"env.num_envs": n, "terrain.type": terrain}
    ov.update(extra or {})
    return T1(load_cfg("T1", ov))


def _actions(n, steps, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [(torch.rand(262144, 12, generator=g) * 1.2 - 0.6)[:n].to(dev) for _ in range(steps)]


def _rollout(env, acts):
    obs, _ = env.reset()
    outs = [obs.clone()]
    for a in acts:
        obs, rew, done, extras = env.step(a)
        outs.append(torch.cat([obs, extras["privileged_obs"], rew[:, None], done[:, None].float(), extras["time_outs"][:, None].float()], dim=1).clone())
    return outs


@pytest.mark.parametrize("n", [1, 33, 100])
def test_ragged_env_counts_match_the_leading_envs_of_a_full_block(n):
    """Environments are independent and RNG streams are keyed by the env index, so with build-time randomisation off the first n envs of
    a 128-env run are BITWISE the n envs of an n-env run (n not a multiple of the 32-env workgroup; n = 1 = half a lane pair's wave)."""
    acts = _actions(128, 12, "cuda:0")
    full = _rollout(_env(128, "plane", NO_DR), acts)  # plane: env origins (which depend on N) do not enter the dynamics
    part = _rollout(_env(n, "plane", NO_DR), [a[:n] for a in acts])
    for a, b in zip(full, part):
        assert torch.equal(a[:n], b)
        assert torch.isfinite(b).all()


def test_full_size_launch_has_the_same_leading_envs():
    """BASELINE-size property: 262,144 envs in one launch; its first 64 envs equal a 64-env run bit for bit, everything stays finite."""
    n = 262144
    acts = _actions(n, 4, "cuda:0")
    big = _rollout(_env(n, "plane", NO_DR), acts)
    small = _rollout(_env(64, "plane", NO_DR), [a[:64] for a in acts])
    for a, b in zip(big, small):
        assert torch.equal(a[:64], b)
        assert torch.isfinite(a).all()


def test_determinism_and_seed_sensitivity():
    acts = _actions(96, 10, "cuda:0")
    a = _rollout(_env(96), acts)
    b = _rollout(_env(96), acts)
    c = _rollout(_env(96, extra={"basic.seed": 43}), acts)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[-1], c[-1])


def test_state_invariants_after_random_rollout():
    n = 512
    env = _env(n, "trimesh")
    env.reset()
    for a in _actions(n, 60, env.device, seed=3):
        obs, rew, done, extras = env.step(a)
    root, q, qd = env.root_states, env.dof_pos, env.dof_vel
    assert torch.allclose(root[:, 3:7].norm(dim=1), torch.ones(n, device=env.device), atol=1e-5)
    lo, hi = env.dof_pos_limits[:, 0], env.dof_pos_limits[:, 1]
    assert ((q > lo - 0.25) & (q < hi + 0.25)).all()  # soft joint limits hold
    assert (qd.abs() <= env.dof_vel_limits + 1e-4).all()  # velocity clamp
    assert (rew >= 0).all()  # only_positive_rewards
    assert torch.isfinite(obs).all() and float(env.episode_stats(reset=False)[-1]) == 0
    # dropped terms (scale 0 in the yaml) are not exported, exported rows are finite
    assert set(extras["rew_terms"].keys()) == set(env.reward_names) and "feet_vel_z" not in extras["rew_terms"]
    ep = env.episode_length_buf
    assert (ep >= 0).all() and (ep <= 61).all()


def test_nonfinite_state_is_reset_and_counted():
    n = 64
    acts = _actions(n, 3, "cuda:0")
    ref = _env(n, extra=NO_DR); ref.reset()
    env = _env(n, extra=NO_DR); env.reset()
    root = env.root_states
    root[5, 7] = float("nan"); root[9, 2] = float("inf")
    env.set_field("root_states", root)
    o1, r1, d1, e1 = env.step(acts[0])
    o0, r0, d0, e0 = ref.step(acts[0])
    assert bool(d1[5]) and bool(d1[9]) and r1[5] == 0 and r1[9] == 0
    assert torch.isfinite(o1).all() and torch.isfinite(env.root_states).all()
    keep = torch.ones(n, dtype=torch.bool, device=env.device); keep[[5, 9]] = False
    assert torch.equal(o1[keep], o0[keep])  # the other envs never noticed
    assert float(env.episode_stats(reset=False)[-1]) == 2.0
    o1, _, _, _ = env.step(acts[1])
    assert torch.isfinite(o1).all()


def test_argument_errors():
    env = _env(8)
    env.reset()
    with pytest.raises(ValueError, match="shape"):
        env.step(torch.zeros(7, 12, device=env.device))
    with pytest.raises(RuntimeError, match="unknown field"):
        env.get_field("no_such_field")
    # host / double / non-contiguous actions are accepted (copied), like the reference accepts any tensor
    out = env.step(torch.zeros(8, 12, dtype=torch.float64))
    assert out[0].shape == (8, 47)
    with pytest.raises(RuntimeError, match="contiguous CUDA"):
        env.step_to(torch.zeros(8, 12, device=env.device), torch.zeros(8, 47), env.privileged_obs_buf, env.rew_buf, env.reset_buf, env.time_out_buf)


@pytest.mark.parametrize("state_dtype", ["fp32", "fp16"])
def test_config5_full_domain_randomisation_16384_envs(state_dtype):
    """BASELINE configs[4]: every randomisation of T1.yaml active (mass / com / friction / latency / kick / push), 16,384 envs, rough
    terrain, state stored in fp32 or fp16.  Kick (cnt % 100) and push (cnt % 250) fall inside the window."""
    n = 16384
    env = _env(n, "trimesh", {"sim.state_dtype": state_dtype})
    env.reset()
    env.common_step_counter = 240
    act = torch.zeros(n, 12, device=env.device)
    dones = 0
    for s in range(20):
        obs, rew, done, extras = env.step(act)
        dones += int(done.sum())
    assert torch.isfinite(obs).all() and torch.isfinite(extras["privileged_obs"]).all()
    push = env.get_field("pushing")
    assert push.abs().max() > 1.0 and push[:, :3].std() > 5.0  # N(0, 10) N forces were drawn at cnt = 250
    delay = env.get_field("delay_steps")[:, 0]
    assert delay.min() == 0 and delay.max() == 9  # latency randomisation over the 10 substeps
    ms = env.get_field("mass_scale")
    assert 0.79 < float(ms[:, 0].min()) < 0.85 and 1.15 < float(ms[:, 0].max()) < 1.21  # base mass x U(0.8, 1.2)
    assert float(env.episode_stats(reset=False)[-1]) == 0 and dones < n // 4


def test_fp16_state_rollout_tracks_the_fp32_rollout():
    """Same seed, same actions, state stored in fp16 vs fp32: after one step the observations differ by no more than the fp16 rounding of
    the initial state allows; over 100 steps of a standing policy the population statistics (reward, episode ends) stay close."""
    n = 4096
    a32, a16 = _env(n, "plane"), _env(n, "plane", {"sim.state_dtype": "fp16"})
    o32, _ = a32.reset()
    o16, _ = a16.reset()
    assert (o32 - o16).abs().max() < 2e-3  # reset noise is identical; only the stored copy is rounded
    act = torch.zeros(n, 12, device=a32.device)
    o32, r32, d32, _ = a32.step(act)
    o16, r16, d16, _ = a16.step(act)
    same = ~(d32 | d16)
    assert same.float().mean() > 0.99
    assert (o32 - o16)[same].abs().max() < 0.05 and (o32 - o16)[same].abs().mean() < 2e-3
    tot32 = tot16 = 0.0
    for _ in range(100):
        _, r32, d32, _ = a32.step(act)
        _, r16, d16, _ = a16.step(act)
        tot32 += float(r32.mean()); tot16 += float(r16.mean())
    s32, s16 = a32.episode_stats(reset=False), a16.episode_stats(reset=False)
    assert float(s16[-1]) == 0 and abs(tot32 - tot16) < 0.05 * abs(tot32) + 0.02
    assert abs(float(s32[0]) - float(s16[0])) <= 0.02 * n + 5  # finished episodes


def test_fallen_robots_come_to_rest_on_their_collision_shapes():
    """Height termination off: robots dropped from 1 m at random orientations (1 to 3 rad from upright) must come to rest ON the ground, carried
    by the trunk box / leg cylinders (the second kernel of the two-kernel scheme), not sink through it.  With contact.body_contacts: false only
    the soles collide and the same robots end up with the trunk far below the ground plane."""
    n = 2048
    over = {"rewards.terminate_height": -10.0, "rewards.terminate_vel": 1.0e9, "rewards.episode_length_s": 1000.0}
    lowest = {}
    for body_contacts in (True, False):
        env = _env(n, "plane", dict(over, **{"contact.body_contacts": body_contacts}))
        env.reset()
        g = torch.Generator(device="cpu").manual_seed(3)
        root = env.root_states.clone().cpu()
        ax = torch.randn(n, 3, generator=g); ax /= ax.norm(dim=1, keepdim=True)
        ang = torch.rand(n, generator=g) * 2.0 + 1.0
        root[:, 2] = 1.0
        root[:, 3:6], root[:, 6] = ax * torch.sin(ang / 2)[:, None], torch.cos(ang / 2)
        root[:, 7:] = 0.0
        env.set_field("root_states", root)
        act = torch.zeros(n, 12, device=env.device)
        for _ in range(300):
            obs, rew, done, extras = env.step(act)
        rs = env.root_states
        assert torch.isfinite(rs).all()
        lowest[body_contacts] = rs[:, 2].clone()
        if body_contacts:
            assert float(env.episode_stats(reset=False)[-1]) == 0
            assert float(rs[:, 2].min()) > 0.005, "a trunk origin sank to the ground plane"
            assert float(rs[:, 7:].abs().mean()) < 0.1, "the fallen robots did not come to rest"
            assert float((extras["rew_terms"]["collision"] < 0).float().mean()) > 0.5  # most of them lie on penalised bodies
    assert float((lowest[False] < 0.0).float().mean()) > 0.5 and float((lowest[True] < 0.0).float().mean()) == 0.0


def test_isaac_layout_state_views_through_the_abi():
    """bg_env_get_state / bg_env_set_state: the Isaac Gym tensor layouts the reference's task code assumes (t1.py:215-220):
    root [N][13], dof [N][12][2] interleaved (pos, vel), net contact force [N][13][3] (only the feet rows 6 and 12 can be non-zero here)."""
    import ctypes as C

    from booster_gym_amd import _lib

    n = 40
    env = _env(n)
    env.reset()
    for a in _actions(n, 25, env.device, seed=5):
        env.step(a * 0.3)
    lib = _lib.load()
    root = torch.empty(n, 13, device=env.device); dof = torch.empty(n, 12, 2, device=env.device); contact = torch.empty(n, 13, 3, device=env.device)
    _lib.check(lib.bg_env_get_state(env._env, _lib.ptr(root), _lib.ptr(dof), _lib.ptr(contact), _lib.current_stream_ptr()))
    assert torch.equal(root, env.root_states) and torch.equal(dof[:, :, 0], env.dof_pos) and torch.equal(dof[:, :, 1], env.dof_vel)
    feet = env.get_field("feet_contact_forces").view(n, 2, 3)
    assert torch.equal(contact[:, 6], feet[:, 0]) and torch.equal(contact[:, 12], feet[:, 1])
    others = [b for b in range(13) if b not in (6, 12)]
    assert contact[:, others].abs().max() == 0 and contact[:, [6, 12], 2].max() > 50.0  # somebody is standing on a foot
    # set_state: write a new root / dof state, read it back through the named fields
    root2 = root.clone(); root2[:, 2] += 0.25; dof2 = dof.clone(); dof2[:, 3, 0] = 0.77
    _lib.check(lib.bg_env_set_state(env._env, _lib.ptr(root2), _lib.ptr(dof2), _lib.current_stream_ptr()))
    assert torch.equal(env.root_states, root2) and torch.allclose(env.dof_pos[:, 3], torch.full((n,), 0.77, device=env.device))
    # model query round-trips the description
    d = _lib.ModelDesc()
    _lib.check(lib.bg_model_get(env._model, C.byref(d)))
    assert d.num_bodies == 13 and d.num_dofs == 12 and abs(d.mass[0] - 19.4304) < 1e-3 and list(d.joint_axis)[1:7] == [2, 1, 3, 2, 2, 1]
    assert lib.bg_env_step_count(env._env) == 25


def _leg_capsule_overlap(env):
    """Deepest overlap [m] between the self-collision capsules of the left and of the right leg, per env, from the env's state alone: forward
    kinematics in torch, 17 sample points along every capsule axis (independent of the kernels and of the oracle)."""
    m, dev = env.model, env.device
    root, q = env.root_states, env.dof_pos
    x, y, z, w = root[:, 3:7].unbind(-1)
    R0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)
    pos = torch.tensor(m.body_pos, dtype=torch.float32, device=dev)

    def rot(ax, a):
        c, s, o, zz = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
        rows = {1: [o, zz, zz, zz, c, -s, zz, s, c], 2: [c, zz, s, zz, o, zz, -s, zz, c], 3: [c, -s, zz, s, c, zz, zz, zz, o]}[ax]
        return torch.stack(rows, dim=-1).view(-1, 3, 3)

    t = torch.linspace(0.0, 1.0, 17, device=dev).view(1, -1, 1)
    pts, rad = [[], []], [[], []]
    caps = m.self_collision_capsules([6, 12])
    for leg in range(2):
        R, p = R0, root[:, 0:3]
        for i in range(6):
            b = 1 + leg * 6 + i
            p = p + (R @ pos[b].view(1, 3, 1)).squeeze(-1)
            R = R @ rot(int(m.joint_axis[b]), q[:, leg * 6 + i])
            for body, a, bb, r in caps[leg]:
                if body == b:
                    A = p + (R @ torch.tensor(a, dtype=torch.float32, device=dev).view(1, 3, 1)).squeeze(-1)
                    B = p + (R @ torch.tensor(bb, dtype=torch.float32, device=dev).view(1, 3, 1)).squeeze(-1)
                    pts[leg].append(A[:, None, :] + t * (B - A)[:, None, :]); rad[leg].append(r)
    depth = torch.full((root.shape[0],), -1.0, device=dev)
    for i in range(2):
        for j in range(2):
            d = torch.cdist(pts[0][i], pts[1][j]).flatten(1).min(dim=1).values
            depth = torch.maximum(depth, rad[0][i] + rad[1][j] - d)
    return depth


def test_self_collision_blocks_leg_interpenetration_at_scale():
    """Size-independent property of the leg-against-leg contacts (reference: self-collision on, envs/T1.yaml:69): 16,384 standing robots are
    told to swing both legs inwards through each other (hip-roll targets far across the mid-plane) for half a second.  With the contacts on, the
    capsules of the two legs end up touching, not overlapping; with asset.self_collisions = 1 (filtered out, as in Isaac Gym) they pass
    through each other.  Depths are measured from the state by an independent torch forward kinematics."""
    n = 16384
    res = {}
    for mask in (0, 1):
        env = _env(n, "plane", {"asset.self_collisions": mask, "rewards.terminate_height": -1.0, "rewards.terminate_vel": 1.0e9, "commands.still_proportion": 1.0})
        env.reset()
        act = torch.zeros(n, 12, device=env.device)
        for _ in range(15):
            env.step(act)
        act[:, 1], act[:, 7] = -0.45, 0.45   # left hip roll negative = inwards
        act[:, 5], act[:, 11] = 0.45, -0.45
        worst = torch.zeros(n, device=env.device)
        for s in range(25):
            env.step(act)
            if s >= 5:
                worst = torch.maximum(worst, _leg_capsule_overlap(env))
        assert torch.isfinite(env.root_states).all() and env.episode_stats(reset=False)[-1].item() == 0
        res[mask] = worst.cpu().numpy()
        del env
    on, off = res[0], res[1]
    print("leg overlap [m] with self-collision on: median %.4f p99 %.4f max %.4f; off: median %.4f" % (np.median(on), np.percentile(on, 99), on.max(), np.median(off)))
    assert np.median(off) > 0.04, np.median(off)           # the command really drives the legs through each other
    assert np.percentile(on, 99) < 0.012 and on.max() < 0.03, (np.percentile(on, 99), on.max())
