"""CPU tests (-m "not gpu"): the C ABI library loads and exports every declared symbol, host-side mirrors of the reference
interface behave like the reference (same errors, same config semantics), the product's per-lane dynamics/RNG headers compiled
for the host agree with the oracle, and the data-parallel path is exact under gloo with world_size 2."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from booster_gym_amd import _lib

    header = open(os.path.join(ROOT, "include", "booster_gym_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(bg_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS), set(declared) ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.bg_version()


def test_ctypes_structs_match_the_c_header(tmp_path):
    from booster_gym_amd import _lib

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "booster_gym_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(bg_env_cfg), '
                   "sizeof(bg_model_desc), offsetof(bg_env_cfg, reward_scale), offsetof(bg_env_cfg, terrain_type), offsetof(bg_env_cfg, noise_gravity), "
                   "offsetof(bg_model_desc, feet_edge_pos), sizeof(bg_param_mirror), offsetof(bg_param_mirror, dst), sizeof(bg_mlp_chain), offsetof(bg_mlp_chain, Y1), "
                   "sizeof(bg_reduce_problem), sizeof(bg_mlp_chain_split), offsetof(bg_mlp_chain_split, X), offsetof(bg_mlp_chain_split, v_out), "
                   "sizeof(bg_mlp_chain_split_bwd), offsetof(bg_mlp_chain_split_bwd, G3), offsetof(bg_mlp_chain_split_bwd, bias_grad1));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    E, M = _lib.EnvCfg, _lib.ModelDesc
    P, Q, R, S, B = _lib.ParamMirror, _lib.MlpChain, _lib.ReduceProblem, _lib.MlpChainSplit, _lib.MlpChainSplitBwd
    assert got == [C.sizeof(E), C.sizeof(M), E.reward_scale.offset, E.terrain_type.offset, E.noise_gravity.offset, M.feet_edge_pos.offset,
                   C.sizeof(P), P.dst.offset, C.sizeof(Q), Q.Y1.offset, C.sizeof(R), C.sizeof(S), S.X.offset, S.v_out.offset, C.sizeof(B), B.G3.offset,
                   B.bias_grad1.offset]


def test_abi_argument_errors_without_gpu():
    """Entry points validate their arguments before touching the device and report through bg_last_error()."""
    from booster_gym_amd import _lib

    lib = _lib.load()
    out = C.c_void_p()
    assert lib.bg_model_create(None, C.byref(out)) < 0 and b"null" in lib.bg_last_error()
    d = _lib.ModelDesc()
    d.num_bodies, d.num_dofs = 5, 4
    assert lib.bg_model_create(C.byref(d), C.byref(out)) < 0 and b"13-body" in lib.bg_last_error()
    assert lib.bg_gae(0, 4, None, None, None, None, None, 0.9, 0.9, None, None, None, None) < 0
    assert lib.bg_env_step(None, None, None) < 0
    # round-3 entry points: descriptors are checked on the host before any launch
    assert lib.bg_mlp_chain_forward_group(None, 1, None) < 0 and b"1 to 4" in lib.bg_last_error()
    q = _lib.MlpChain(); q.M, q.K0, q.N1, q.N2, q.N3 = 128, 64, 256, 128, 128
    assert lib.bg_mlp_chain_forward_group(C.addressof(q), 1, None) == -1 and b"bad argument" in lib.bg_last_error()
    assert lib.bg_mlp_chain_forward_group(C.addressof(q), 5, None) < 0
    assert lib.bg_critic_values_gae(0, 4, None, None, None, None, None, None, 0.9, 0.9, None, None, None, None, None, None) < 0
    assert lib.bg_reduce_group(None, 1, None) < 0
    m1 = _lib.ParamMirror(0, 4, 4, 0, 2, 0, None)  # ld < cols and no destination
    one = (C.c_float * 16)()
    assert lib.bg_optimizer_step(16, one, one, one, one, one, 1, 0.9, 0.999, 1e-8, 1.0, None, 0, 0, None, None, None, 0, 0, 1.0, 0.01, 1e-5, 1e-2, one,
                                 C.addressof(m1), 1, None) < 0 and b"mirror" in lib.bg_last_error()
    # round-4 entry points
    assert lib.bg_mlp_weight_grad_group_partial(None, 1, None) < 0 and b"bg_mlp_weight_grad_group_partial" in lib.bg_last_error()
    tail = lambda **kw: lib.bg_update_tail(kw.get("wg"), kw.get("nwg", 0), kw.get("rd"), kw.get("nrd", 0), 16, one, one, one, one, one, kw.get("step", 1), 0.9, 0.999,
                                           1e-8, 1.0, None, 0, 0, None, None, None, 0, 0, 1.0, 0.01, 1e-5, 1e-2, kw.get("sync", one), one,
                                           kw.get("mir"), kw.get("nmir", 0), None)
    assert tail(sync=None) == -1 and b"bg_update_tail: bad argument" in lib.bg_last_error()
    assert tail(step=0) == -1
    assert tail(nrd=9) == -1 and b"at most 8 reductions" in lib.bg_last_error()
    assert tail(mir=C.addressof(m1), nmir=1) == -1 and b"mirror" in lib.bg_last_error()
    rp = _lib.ReduceProblem()  # an empty descriptor
    assert tail(rd=(_lib.ReduceProblem * 1)(rp), nrd=1) == -1 and b"bad reduction descriptor" in lib.bg_last_error()
    wp = (_lib.WgradProblem * 1)()  # an empty weight-gradient problem: refused by the shared descriptor check, under this entry's name
    assert tail(wg=wp, nwg=1) < 0 and b"bg_update_tail" in lib.bg_last_error()
    if not torch.cuda.is_available():
        from booster_gym_amd.utils.urdf import FlatModel

        # a valid model is accepted, but an env cannot be created without a HIP device: no CPU path
        m = FlatModel.load(os.path.join(ROOT, "booster_gym_amd", "resources", "T1", "T1_locomotion.flat.json"))
        d = _lib.ModelDesc(); d.num_bodies, d.num_dofs = 13, 12
        for b in range(13):
            d.parent[b], d.joint_axis[b], d.mass[b] = int(m.parent[b]), int(m.joint_axis[b]), float(m.mass[b])
        assert lib.bg_model_create(C.byref(d), C.byref(out)) == 0
        cfg = _lib.EnvCfg(); cfg.num_envs, cfg.decimation, cfg.sim_dt = 4, 10, 0.002
        env = C.c_void_p()
        assert lib.bg_env_create(C.byref(cfg), out, C.byref(env)) < 0 and b"no HIP device" in lib.bg_last_error()
        lib.bg_model_destroy(out)


# ------------------------------------------------------------------ host mirrors of the reference interface
def test_policy_io_contract_matches_the_deployed_configuration(flat_model):
    """The exported actor is consumed by deploy/utils/policy.py:34-73 under deploy/configs/T1.yaml (numbers-only fixture: tests/golden/deploy_contract.json).
    What a policy trained HERE must agree with for that consumer to work: 47 observations / 12 actions, 500 Hz x decimation 10, action scale and clip,
    the observation scales, the default pose the actions are offsets from, and the leg joints' proportional gains.  Two differences are the reference's
    own (its training yaml against its deploy yaml), asserted so that they stay visible: the ankles' derivative gain (1 in training, envs/T1.yaml:93; 3 on
    the robot) and two torque limits (hip pitch: URDF effort 45, 60 on the robot; hip roll: 30 / 25)."""
    from booster_gym_amd.utils.config import load_cfg

    dep = json.load(open(os.path.join(ROOT, "tests", "golden", "deploy_contract.json")))
    cfg = load_cfg("T1", {})
    assert cfg["env"]["num_observations"] == dep["num_observations"] and cfg["env"]["num_actions"] == dep["num_actions"]
    assert cfg["sim"]["dt"] == dep["dt"] and cfg["control"]["decimation"] == dep["decimation"] and cfg["control"]["action_scale"] == dep["action_scale"]
    nz = cfg["normalization"]
    for k, v in dep["normalization"].items():
        assert float(nz[k]) == float(v), k
    lo, hi = cfg["commands"]["gait_frequency"]
    assert lo <= dep["gait_frequency"] <= hi  # the robot walks at a frequency the policy was trained on
    names = flat_model.dof_names
    pick = lambda table, name: [v for k, v in table.items() if k in name][-1]
    kp = [pick(cfg["control"]["stiffness"], n) for n in names]
    kd = [pick(cfg["control"]["damping"], n) for n in names]
    dja = cfg["init_state"]["default_joint_angles"]
    q0 = [([v for k, v in dja.items() if k != "default" and k in n] or [dja["default"]])[-1] for n in names]
    assert kp == dep["leg_stiffness"] and q0 == dep["leg_default_qpos"]
    ankle = ["Ankle" in n for n in names]
    assert [d for d, a in zip(kd, ankle) if not a] == [d for d, a in zip(dep["leg_damping"], ankle) if not a]
    assert {d for d, a in zip(kd, ankle) if a} == {1} and {d for d, a in zip(dep["leg_damping"], ankle) if a} == {3}
    diff = {n: (e, t) for n, e, t in zip(names, flat_model.dof_effort.tolist(), dep["leg_torque_limit"]) if e != t}
    assert diff == {"Left_Hip_Pitch": (45.0, 60), "Right_Hip_Pitch": (45.0, 60), "Left_Hip_Roll": (30.0, 25), "Right_Hip_Roll": (30.0, 25)}


def test_config_surface_is_a_superset_of_the_reference_yaml():
    from booster_gym_amd.utils.config import load_cfg

    cfg = load_cfg("T1")
    ref_path = "/root/reference/envs/T1.yaml"
    if os.path.isfile(ref_path):
        ref = yaml.safe_load(open(ref_path))

        def walk(a, b, path=""):
            for k, v in b.items():
                assert k in a, path + str(k)
                if isinstance(v, dict):
                    walk(a[k], v, path + str(k) + ".")
                elif path + str(k) not in ("basic.headless", "runner.use_wandb", "viewer.record_video", "sim.physics_engine"):
                    assert a[k] == v, (path + str(k), a[k], v)
        walk(cfg, ref)
    assert cfg["env"]["num_envs"] == 4096 and cfg["runner"]["horizon_length"] == 24 and cfg["control"]["decimation"] == 10
    assert isinstance(cfg["contact"]["stiffness"], float)
    assert load_cfg("T1", {"env.num_envs": 8})["env"]["num_envs"] == 8
    with pytest.raises(FileNotFoundError):
        load_cfg("NoSuchTask")


def test_runner_cli_overrides_follow_the_reference():
    """runner.py:44-68: 8 flags; num_envs lands in cfg['env'], everything else in cfg['basic']; --task is required."""
    from booster_gym_amd.utils.runner import Runner

    r = object.__new__(Runner)
    r.test = False
    r._get_args(["--task", "T1", "--num_envs", "16", "--seed", "7", "--max_iterations", "3", "--sim_device", "cuda:1", "--headless", "x"])
    r._update_cfg_from_args()
    assert r.cfg["env"]["num_envs"] == 16 and r.cfg["basic"]["seed"] == 7 and r.cfg["basic"]["max_iterations"] == 3
    assert r.cfg["basic"]["sim_device"] == "cuda:1" and r.cfg["basic"]["headless"] is True and r.cfg["basic"]["task"] == "T1"
    assert r.cfg["viewer"]["record_video"] is False
    with pytest.raises(SystemExit):
        r._get_args([])


def test_env_refuses_cpu_and_bad_configs():
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    with pytest.raises(ValueError, match="GPU only"):
        T1(load_cfg("T1", {"basic.sim_device": "cpu", "env.num_envs": 4}))
    with pytest.raises(ValueError, match="Invalid terrain type"):
        T1(load_cfg("T1", {"terrain.type": "moon", "env.num_envs": 4}))
    with pytest.raises(ValueError, match="up-axis"):
        T1(load_cfg("T1", {"sim.up_axis": "y", "env.num_envs": 4}))


def test_apply_randomization_matches_reference_and_raises_like_it():
    from booster_gym_amd.utils.utils import apply_randomization, rand_spec

    d = np.load(os.path.join(HERE, "golden", "apply_randomization.npz"))
    x = torch.tensor(d["x"])
    for dist in ("gaussian", "uniform"):
        for op in ("additive", "scaling"):
            torch.manual_seed(7)
            y, noise = apply_randomization(x, {"distribution": dist, "operation": op, "range": [0.3, 0.7]}, return_noise=True)
            assert torch.allclose(y, torch.tensor(d[f"{dist}_{op}_y"])) and torch.allclose(noise, torch.tensor(d[f"{dist}_{op}_noise"]))
    assert apply_randomization(x, None) is x
    with pytest.raises(ValueError, match="distribution"):
        apply_randomization(x, {"distribution": "cauchy", "operation": "additive", "range": [0, 1]})
    with pytest.raises(ValueError, match="operation"):
        apply_randomization(x, {"distribution": "uniform", "operation": "xor", "range": [0, 1]})
    assert isinstance(apply_randomization(1.0, {"distribution": "uniform", "operation": "scaling", "range": [2.0, 2.0]}), float)
    assert rand_spec(None) == (0, 0.0, 0.0) and rand_spec({"distribution": "gaussian", "operation": "scaling", "range": [1, 2]}) == (2, 1.0, 2.0)
    assert rand_spec({"distribution": "uniform", "operation": "additive", "range": [1, 2]})[0] == 3


def test_terrain_generation_and_height_query():
    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.terrain import Terrain
    from oracle.task_ref import terrain_heights

    cfg = load_cfg("T1")
    t = Terrain("cpu", cfg["terrain"], seed=42)
    hf = t.height_field_raw
    assert hf.shape == (900, 200) and hf.dtype == np.int16  # 8 x 100 px + 2 x 50 px border (terrain.py:38-45)
    assert (hf[:50] == 0).all() and (hf[:, :50] == 0).all() and (hf[-50:] == 0).all()  # flat border
    rough, obst = hf[50:450, 50:150], hf[450:850, 50:150]
    assert np.abs(rough).max() <= 10 and rough.std() > 1  # +-0.05 m / 0.005
    assert set(np.unique(obst)).issubset({-4, -2, 0, 2, 4})  # +-0.02 m, +-0.01 m obstacles
    assert (obst[35:65, 35:65] == 0).all()  # 3 m flat platform of the first obstacle strip
    assert (Terrain("cpu", cfg["terrain"], seed=42).height_field_raw == hf).all() and (Terrain("cpu", cfg["terrain"], seed=43).height_field_raw != hf).any()
    xy = np.random.default_rng(0).uniform([0, 0], [80, 10], (200, 2))
    tdict = dict(height_field_raw=hf, hscale=0.1, vscale=0.005, border_px=50)
    assert np.allclose(t.terrain_heights(torch.tensor(np.c_[xy, np.zeros(200)])).numpy(), terrain_heights(tdict, xy), atol=1e-6)
    plane = Terrain("cpu", dict(cfg["terrain"], type="plane"))
    assert plane.terrain_heights(torch.zeros(5, 3)).abs().max() == 0
    with pytest.raises(ValueError):
        Terrain("cpu", dict(cfg["terrain"], type="lava"))


def test_urdf_loader_collapses_fixed_joints(tmp_path):
    from booster_gym_amd.utils.urdf import FlatModel, load_urdf

    urdf = """<robot name="toy"><link name="a"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="2"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial>
    <collision><origin xyz="0 0 0" rpy="0 0 0"/><geometry><box size="1 1 1"/></geometry></collision></link>
    <link name="b"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="2"/><inertia ixx="1" ixy="0" ixz="0" iyy="2" iyz="0" izz="3"/></inertial></link>
    <link name="c"><inertial><origin xyz="0 0 -0.1" rpy="0 0 0"/><mass value="1"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial></link>
    <joint name="fix" type="fixed"><origin xyz="1 0 0" rpy="0 0 1.5707963267948966"/><parent link="a"/><child link="b"/></joint>
    <joint name="hinge" type="revolute"><origin xyz="0 0.5 0" rpy="0 0 0"/><parent link="a"/><child link="c"/><axis xyz="0 1 0"/>
    <limit lower="-1" upper="2" effort="5" velocity="3"/></joint></robot>"""
    p = tmp_path / "toy.urdf"
    p.write_text(urdf)
    m = load_urdf(str(p))
    assert m.body_names == ["a", "c"] and m.joint_axis.tolist() == [0, 2] and m.parent.tolist() == [-1, 0]
    assert np.isclose(m.mass[0], 4.0) and np.allclose(m.com[0], [0.5, 0, 0])
    # b's inertia rotated 90 deg about z (ixx<->iyy), both shifted 0.5 m along x: Iyy/Izz += m d^2
    assert np.allclose(m.inertia[0][:3], [1 + 2, 1 + 1 + 2 * 2 * 0.25, 1 + 3 + 2 * 2 * 0.25])
    assert m.dof_lower.tolist() == [-1.0] and m.dof_effort.tolist() == [5.0] and m.shapes[0]["type"] == "box"
    m.save(str(tmp_path / "toy.json"))
    m2 = FlatModel.load(str(tmp_path / "toy.json"))
    assert m2.body_names == m.body_names and np.allclose(m2.inertia, m.inertia)
    bad = urdf.replace('<axis xyz="0 1 0"/>', '<axis xyz="0.7 0.7 0"/>')
    p.write_text(bad)
    with pytest.raises(ValueError, match="axis"):
        load_urdf(str(p))


def test_experience_buffer_and_recorder(tmp_path):
    from booster_gym_amd.utils.buffer import ExperienceBuffer
    from booster_gym_amd.utils.recorder import Recorder

    b = ExperienceBuffer(4, 3, "cpu")
    b.add_buffer("obses", (5,), extra_rows=1); b.add_buffer("dones", (), dtype=torch.bool)
    assert b["obses"].shape == (5, 3, 5) and b["dones"].shape == (4, 3) and b["dones"].dtype == torch.bool and b.names() == ("obses", "dones")
    b.row("obses", 4).fill_(2.0)  # the carried row: observation after the last step
    b.row("obses", 2).copy_(torch.ones(3, 5))
    assert b.flat("obses").shape == (12, 5) and b.flat("obses", with_carry=True).shape == (15, 5) and b.flat("obses")[6:9].sum() == 15
    assert b.flat("obses").data_ptr() == b["obses"].data_ptr()  # views, not copies
    b.roll()
    assert b["obses"][0].eq(2.0).all() and b.nbytes() == 5 * 3 * 5 * 4 + 4 * 3
    with pytest.raises(KeyError):
        b.add_buffer("obses", (5,))
    rec = Recorder({"basic": {"task": "T1"}, "runner": {"use_wandb": False}}, root=str(tmp_path))
    rec.record_statistics({"value_loss": 1.5, "lr": 1e-5}, 3)
    path = rec.save({"model": {}, "optimizer": {}, "curriculum": torch.zeros(21, 21)}, 100)
    assert path.endswith(os.path.join("nn", "model_100.pth")) and os.path.isfile(path)
    assert os.path.isfile(os.path.join(rec.dir, "config.yaml"))
    lines = open(os.path.join(rec.dir, "summaries", "scalars.jsonl")).read().strip().splitlines()
    assert len(lines) == 2 and '"value_loss"' in lines[0]
    assert set(torch.load(path, weights_only=True).keys()) == {"model", "optimizer", "curriculum"}


def test_model_state_dict_keys_match_the_reference_layout():
    from booster_gym_amd.utils.model import ActorCritic

    m = ActorCritic(12, 47, 14)
    keys = list(m.state_dict().keys())
    assert keys == ["logstd"] + [f"critic.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")] + [f"actor.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    assert sum(p.numel() for p in m.parameters()) == 177945
    assert m.act(torch.zeros(3, 47)).scale[0, 0].item() == pytest.approx(np.exp(-2.0))
    assert m.est_value(torch.zeros(3, 47), torch.zeros(3, 14)).shape == (3,)
    # the reference's exported actor (deploy/models/T1.pt) has exactly the actor's tensor shapes
    W = np.load(os.path.join(HERE, "golden", "t1_actor.npz"))
    for i in (0, 2, 4, 6):
        assert tuple(W[f"{i}.weight"].shape) == tuple(m.actor[i].weight.shape)


def test_observation_layout_equals_what_the_reference_deploy_code_builds(flat_model):
    """tests/golden/deploy_policy.npz holds what the reference's OWN deploy-side class computes (deploy/utils/policy.py:34-73 run on 96 seeded robot states
    with its yaml and its trained TorchScript actor; tests/golden/make_policy_fixture.py).  The training-side observation of this build -- the oracle's
    compute_observations, which the HIP env step is held to entry by entry in tests/test_gpu_env.py -- must BE that vector on the same state: layout,
    scales, default pose, the gait-gated phase entries, the previous (clipped) actions; then the clip and the joint targets the class derives."""
    import oracle.task_ref as tr
    from booster_gym_amd.utils.config import load_cfg

    d = np.load(os.path.join(ROOT, "tests", "golden", "deploy_policy.npz"))
    cfg = load_cfg("T1", {})
    names = flat_model.dof_names
    dja = cfg["init_state"]["default_joint_angles"]
    default = np.array([([v for k, v in dja.items() if k != "default" and k in n] or [dja["default"]])[-1] for n in names], dtype=np.float64)
    assert np.array_equal(default.astype(np.float32), d["default_qpos"][11:])  # robot joints 11..22 = the 12 policy DoFs (policy.py:60)
    nz = cfg["normalization"]
    norm = dict(gravity=nz["gravity"], lin_vel=nz["lin_vel"], ang_vel=nz["ang_vel"], dof_pos=nz["dof_pos"], dof_vel=nz["dof_vel"], push_force=0.1, push_torque=0.5)
    E, T = d["obs"].shape[:2]
    prev = np.concatenate([np.zeros((E, 1, 12), np.float32), d["actions"][:, :-1]], axis=1)  # obs[35:47] = the actions of the call before (policy.py:62)
    flat = lambda a: a.reshape(E * T, *a.shape[2:]).astype(np.float64)
    K = E * T
    s = dict(projected_gravity=flat(d["projected_gravity"]), base_ang_vel=flat(d["base_ang_vel"]), commands=flat(d["smoothed_commands"]),
             gait_frequency=flat(d["gait_frequency"]), gait_process=flat(d["gait_process"]), dof_pos=flat(d["dof_pos"])[:, 11:], dof_vel=flat(d["dof_vel"])[:, 11:],
             actions=flat(prev), root_states=np.zeros((K, 13)), base_mass_scaled=np.zeros((K, 4)), base_lin_vel=np.zeros((K, 3)), push_force=np.zeros((K, 3)),
             push_torque=np.zeros((K, 3)))
    obs, _ = tr.compute_observations(s, norm, default, None, noisy=None)
    ref = flat(d["obs"])
    closed = s["gait_frequency"] == 0
    assert 10 <= closed.sum() < K  # both states of the gait gate are in the fixture
    # the deploy code also multiplies the commands by the gate (policy.py:49-57); the env does not need to: its standing envs carry commands of exactly
    # zero (t1.py:381-386), and here the gate closes below |smoothed command| = 1e-5
    assert np.abs(obs[closed, 6:9]).max() < 1e-5 and np.all(ref[closed, 6:11] == 0)
    cols = np.r_[0:6, 9:47]
    assert np.allclose(obs[:, cols], ref[:, cols], rtol=0, atol=2e-6), np.abs(obs[:, cols] - ref[:, cols]).max()
    assert np.allclose(obs[~closed, 6:9], ref[~closed, 6:9], rtol=0, atol=1e-7)
    # what the class does with the network's output: clip to +-clip_actions, targets = default pose + action_scale * actions on the 12 leg joints
    clip = float(nz["clip_actions"])
    assert np.abs(d["raw_actions"]).max() > clip  # (the clip is exercised)
    assert np.array_equal(np.clip(d["raw_actions"], -clip, clip), d["actions"])
    tgt = np.tile(d["default_qpos"], (E, T, 1)); tgt[..., 11:] += np.float32(cfg["control"]["action_scale"]) * d["actions"]
    assert np.array_equal(tgt, d["dof_targets"])
    # the smoothing of policy.py:39-40: the commands move towards their set point by at most dt * decimation per call
    step = np.abs(np.diff(d["smoothed_commands"], axis=1)).max()
    assert step <= float(d["policy_interval"]) * (1 + 1e-6) and abs(float(d["policy_interval"]) - cfg["sim"]["dt"] * cfg["control"]["decimation"]) < 1e-12


# ------------------------------------------------------------------ product headers compiled for the host vs the oracle
@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("hh")
    libs = {}
    for name in ("harness", "rng_harness"):
        so = str(d / f"lib{name}.so")
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared"] + (SAN_FLAGS if SANITIZE else ["-O2"]) + ["-o", so, os.path.join(HERE, "host_harness", f"{name}.cpp")])
        libs[name] = C.CDLL(so)
    return libs


# BG_SANITIZE=1 (set by tests/test_sanitizers.py, which re-runs the host-code tests of this file in a child process under LD_PRELOAD=libasan): every
# piece of native host code these tests touch is built with the address and undefined-behaviour sanitizers -- the harnesses above, the oracle's C
# source (oracle/dyn_ref.py honours the same variable) and the URDF loader + model object of the product as a stand-alone library (below)
SANITIZE = os.environ.get("BG_SANITIZE", "0") == "1"
SAN_FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
_san_urdf_lib = None


def _urdf_lib():
    """The library that exports bg_model_load_urdf: the product's, or (BG_SANITIZE) csrc/bg_urdf.cpp + csrc/bg_model.cpp built by g++ with sanitizers."""
    global _san_urdf_lib
    from booster_gym_amd import _lib

    if not SANITIZE:
        return _lib.load()
    if _san_urdf_lib is None:
        import tempfile

        so = os.path.join(tempfile.mkdtemp(prefix="bg_san_"), "libbgurdf_san.so")
        src = os.path.join(ROOT, "booster_gym_amd", "csrc")
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared"] + SAN_FLAGS + ["-o", so, os.path.join(src, "bg_urdf.cpp"), os.path.join(src, "bg_model.cpp")])
        l = C.CDLL(so)
        l.bg_last_error.restype = C.c_char_p
        l.bg_model_body_name.restype = C.c_char_p
        l.bg_model_dof_name.restype = C.c_char_p
        l.bg_model_body_name.argtypes = [C.c_void_p, C.c_int32]
        l.bg_model_dof_name.argtypes = [C.c_void_p, C.c_int32]
        l.bg_model_find_body.argtypes = [C.c_void_p, C.c_char_p]
        l.bg_model_load_urdf.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        l.bg_model_get.argtypes = [C.c_void_p, C.c_void_p]
        l.bg_model_destroy.argtypes = [C.c_void_p]
        _san_urdf_lib = l
    return _san_urdf_lib


def test_product_rng_header_matches_oracle_philox(harness):
    from oracle.task_ref import rand4

    hh = harness["rng_harness"]
    hh.hh_rand4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    for seed, env, step, stream in [(42 | (1 << 32), 0, 0, 0), (7 | (3 << 32), 4095, 123456, 25), (0xFFFFFFFF, 0xFFFFFFFF, 99, 84)]:
        u, n = np.zeros(4, np.float32), np.zeros(4, np.float32)
        hh.hh_rand4(seed, env, step, stream, u.ctypes.data, n.ctypes.data)
        ur, nr = rand4(seed, np.array([env], dtype=np.uint32), step, stream)
        assert np.array_equal(u, ur[0]) and np.allclose(n, nr[0], atol=2e-6)


def test_product_dynamics_header_matches_oracle(harness, flat_model):
    """The per-lane fp32 articulated-body code of the HIP kernels, compiled for the host, against the float64 oracle: one leg per lane (bg_dyn.h)
    and, the same generic code with two-wide scalars, one env per lane (bg_dyn_pk.h)."""
    from oracle.dyn_ref import DEFAULT_PHYS, DynRef

    hh, m = harness["harness"], flat_model

    class ModelDev(C.Structure):
        _fields_ = [("pos", C.c_float * 3 * 13), ("mass", C.c_float * 13), ("com", C.c_float * 3 * 13), ("inertia", C.c_float * 6 * 13), ("q_lo", C.c_float * 12),
                    ("q_hi", C.c_float * 12), ("qd_max", C.c_float * 12), ("tau_lim", C.c_float * 12), ("corner", C.c_float * 3 * 4),
                    ("sph_n", C.c_int), ("sph_first", C.c_int * 13), ("sph_cnt", C.c_int * 13), ("sph_pos", C.c_float * 3 * 16), ("sph_r", C.c_float * 16),
                    ("cap_c", C.c_float * 3 * 2 * 2), ("cap_h", C.c_float * 2 * 2), ("cap_r", C.c_float * 2 * 2)]

    class Cfg(C.Structure):
        _fields_ = [("dt", C.c_float), ("g", C.c_float * 3)] + [(k, C.c_float) for k in ("contact_k", "contact_d", "contact_ramp", "friction_visc", "limit_k", "limit_d",
                                                                                      "terrain_mu", "terrain_restitution")] + [("clamp_qd", C.c_int), ("body_gate", C.c_float), ("self_on", C.c_int)] + \
                   [(k, C.c_float) for k in ("self_k", "self_d", "self_mu", "self_visc")] + [("zmask", C.c_int)]

    class Terr(C.Structure):
        _fields_ = [("type", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("border_px", C.c_int), ("inv_hscale", C.c_float), ("vscale", C.c_float), ("hf", C.c_void_p)]

    md = ModelDev()
    for b in range(13):
        md.mass[b] = m.mass[b]
        for a in range(3):
            md.pos[b][a], md.com[b][a] = m.body_pos[b, a], m.com[b, a]
        for a in range(6):
            md.inertia[b][a] = m.inertia[b, a]
    for j in range(12):
        md.q_lo[j], md.q_hi[j], md.qd_max[j], md.tau_lim[j] = m.dof_lower[j], m.dof_upper[j], m.dof_velocity[j], m.dof_effort[j]
    corners = [[0.1215, 0.05, -0.03], [0.1215, -0.05, -0.03], [-0.1015, 0.05, -0.03], [-0.1015, -0.05, -0.03]]
    for k in range(4):
        for a in range(3):
            md.corner[k][a] = corners[k][a]
    sph = m.contact_spheres(exclude_bodies=[6, 12])  # the product's own derivation of the contact spheres (utils/urdf.py)
    md.sph_n = len(sph)
    for k, (b, c, r) in enumerate(sph):
        if md.sph_cnt[b] == 0:
            md.sph_first[b] = k
        md.sph_cnt[b] += 1
        md.sph_r[k] = r
        for a in range(3):
            md.sph_pos[k][a] = c[a]
    assert len(sph) == 16 and [md.sph_cnt[b] for b in (0, 3, 4, 9, 10)] == [8, 2, 2, 2, 2]
    for leg, caps in enumerate(m.self_collision_capsules([6, 12])):  # the product's own derivation of the self-collision capsules
        for k, (body, a, b, r) in enumerate(caps):
            assert body == (4, 6)[k] + 6 * leg
            ax = (2, 0)[k]
            md.cap_h[leg][k], md.cap_r[leg][k] = 0.5 * (b[ax] - a[ax]), r
            for i in range(3):
                md.cap_c[leg][k][i] = 0.5 * (a[i] + b[i])
    cfg = Cfg(); cfg.dt = DEFAULT_PHYS["dt"]; cfg.clamp_qd = 1; cfg.body_gate = DEFAULT_PHYS["body_gate_height"]
    cfg.self_on = 1
    for k in ("self_k", "self_d", "self_mu", "self_visc"):
        setattr(cfg, k, DEFAULT_PHYS[k])
    # links whose origin lies on the parent's z axis in both legs take the kernels' specialised code (Phys::zmask); the T1: hip roll, hip yaw, ankle pitch
    cfg.zmask = sum(1 << i for i in range(6) if all(m.body_pos[1 + 6 * leg + i, 0] == 0 and m.body_pos[1 + 6 * leg + i, 1] == 0 for leg in range(2)))
    assert cfg.zmask == 0b010110
    for a in range(3):
        cfg.g[a] = DEFAULT_PHYS["g"][a]
    for k in ("contact_k", "contact_d", "contact_ramp", "friction_visc", "limit_k", "limit_d", "terrain_mu", "terrain_restitution"):
        setattr(cfg, k, DEFAULT_PHYS[k])
    rng = np.random.default_rng(1)
    hf = rng.integers(-10, 10, size=(60, 60)).astype(np.int16)
    # contact: False = airborne, True = standing height (sole contacts), "low" = trunk 0.15-0.5 m above the ground in any orientation, so that
    # the trunk box and the hip-yaw / shank cylinders touch (explicit sphere contacts) as well
    # "crossed": airborne with the hip rolls drawn inwards, so that shanks / feet of the two legs overlap (leg-against-leg contacts)
    for terrain, contact, tol in ((None, False, 2e-5), (None, True, 5e-4), (dict(height_field_raw=hf, hscale=0.1, vscale=0.005, border_px=30), True, 5e-4),
                                  (None, "low", 1e-3), (dict(height_field_raw=hf, hscale=0.1, vscale=0.005, border_px=30), "low", 1e-3),
                                  (None, "crossed", 5e-4)):
        d = DynRef(m, terrain=terrain)
        t = Terr()
        if terrain is not None:
            t.type, t.rows, t.cols, t.border_px, t.inv_hscale, t.vscale, t.hf = 1, 60, 60, 30, 10.0, 0.005, hf.ctypes.data
        worst, worst_pk, ncontact, nbody, crossed = 0.0, 0.0, 0, 0, contact == "crossed"
        if crossed:
            contact = False
        for _ in range(150):
            root = np.zeros(13); root[2] = (rng.uniform(0.15, 0.5) if contact == "low" else rng.uniform(0.55, 0.72)) if contact else 5.0
            root[:2] = rng.uniform(-1, 1, 2)
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 3.0 if contact == "low" else 0.3)
            root[3:6], root[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
            root[7:13] = rng.normal(size=6) * (0.3 if contact else 1.0)
            q = (np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2) + rng.normal(size=12) * 0.1) if contact else rng.uniform(m.dof_lower - 0.05, m.dof_upper + 0.05)
            if crossed:
                q[[1, 7]], q[[0, 6]], q[[3, 9]] = (rng.uniform(-0.3, 0.0), rng.uniform(0.0, 0.3)), rng.uniform(-0.6, 0.2, 2), rng.uniform(0.0, 0.8, 2)
            elif not contact:  # plain airborne case: smooth dynamics only, legs apart
                q[[1, 7]] = rng.uniform(0.2, 1.0), rng.uniform(-1.0, -0.2)
            qd, tau, w = rng.normal(size=12), rng.uniform(-m.dof_effort, m.dof_effort), rng.normal(size=6) * 10
            ms, co = rng.uniform(0.8, 1.2, 13), rng.uniform(-0.05, 0.05, (13, 3))
            fm = np.array([rng.uniform(0.1, 2), rng.uniform(0.5, 1.5), rng.uniform(0.1, 0.9)] * 2)
            f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
            arrs = [f32(ms), f32(co.reshape(39)), f32(fm), f32(root), f32(q), f32(qd), f32(tau), f32(w)]
            qa, cf32, bcf = np.zeros(18, np.float32), np.zeros(6, np.float32), np.zeros((13, 3), np.float32)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            hh.hh_forward(C.byref(md), C.byref(cfg), C.byref(t), *[p(a) for a in arrs], p(qa), p(cf32), 0, p(bcf))
            qacc, cf = d.forward(arrs[3], arrs[4], arrs[5], arrs[6], base_wrench=arrs[7], mass_scale=arrs[0], com_off=arrs[1].reshape(13, 3), foot_mat=arrs[2])
            tol_s = tol if np.abs(cf).max() == 0 else max(tol, 5e-4)  # any contact (also a chance leg-against-leg one in the airborne case) is stiff
            worst = max(worst, np.abs(qa - qacc).max() / max(1.0, np.abs(qacc).max()) / tol_s)
            if contact != "low":  # the packed lane code (one env per lane, bg_dyn_pk.h) carries no non-foot body contacts: every other case
                qp, cfp = np.zeros(18, np.float32), np.zeros(6, np.float32)
                hh.hh_forward_pk(C.byref(md), C.byref(cfg), C.byref(t), *[p(a) for a in arrs], p(qp), p(cfp))
                worst_pk = max(worst_pk, np.abs(qp - qacc).max() / max(1.0, np.abs(qacc).max()) / tol_s)
                assert np.abs(cfp.reshape(2, 3) - cf[[6, 12]]).max() <= 2e-3 * max(1.0, np.abs(cf).max())
            ncontact += int(np.abs(cf).max() > 0)
            body_rows = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11]
            nbody += int(np.abs(cf[body_rows]).max() > 0)
            assert np.abs(bcf[body_rows] - cf[body_rows]).max() <= 2e-3 * max(1.0, np.abs(cf).max())
        assert worst < 1.0, (terrain is not None, contact, crossed, worst)
        assert worst_pk < 1.0, ("packed", terrain is not None, contact, crossed, worst_pk)
        assert (ncontact > 50) == bool(contact or crossed), ncontact
        assert (nbody > 30) == (contact == "low" or crossed), nbody  # crossed: the shank rows carry leg-against-leg forces


# ------------------------------------------------------------------ data parallel under gloo, world_size 2
def _dp_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BG_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F

    from booster_gym_amd.utils.model import ActorCritic
    from booster_gym_amd.utils.parallel import DataParallel
    from oracle.ppo_ref import discount_values, surrogate_loss

    torch.set_num_threads(1)
    dp = DataParallel()
    d = np.load(os.path.join(HERE, "golden", "ppo_epoch.npz"))
    model = ActorCritic(12, 47, 14)
    if rank == 0:
        model.load_state_dict({k[3:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd_")})
    dp.broadcast_parameters(model)  # rank 1 starts from different random weights: the broadcast must fix that
    N = d["obses"].shape[1]
    sl = slice(rank * N // world, (rank + 1) * N // world)  # shard the ENVIRONMENTS
    t = lambda k: torch.tensor(d[k])
    obs, priv, act = t("obses")[:, sl], t("priv")[:, sl], t("actions")[:, sl]
    rew, dones, touts = t("rewards")[:, sl].clone(), t("dones")[:, sl], t("time_outs")[:, sl]
    vals = model.est_value(obs, priv)
    lastv = model.est_value(t("last_obs")[sl], t("last_priv")[sl])
    with torch.no_grad():
        rew[touts] = vals[touts]
        adv = discount_values(rew, dones | touts, vals, lastv, 0.995, 0.95)
        ret = vals + adv
        sums = torch.tensor([adv.double().sum(), adv.double().square().sum(), float(adv.numel())], dtype=torch.float64)
        dp.sum_(sums)  # exchange (1): global advantage moments
        mean = sums[0] / sums[2]
        std = torch.sqrt((sums[1] - sums[2] * mean * mean) / (sums[2] - 1.0))
        advn = ((adv.double() - mean) / (std + 1e-8)).float()
    dist = model.act(obs)
    logp = dist.log_prob(act).sum(-1)
    loss = F.mse_loss(vals, ret) + surrogate_loss(t("old_logp")[:, sl], logp, advn)
    loss = loss + torch.clip(dist.loc - 1.0, min=0.0).square().mean() + torch.clip(dist.loc + 1.0, max=0.0).square().mean() - 0.01 * dist.entropy().sum(-1).mean()
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    dp.average_(flat)  # exchange (2): one flat gradient bucket
    kl = torch.sum(torch.log(dist.scale / torch.exp(t("old_logstd"))) + 0.5 * (torch.exp(t("old_logstd")) ** 2 + (dist.loc - t("old_mu")[:, sl]) ** 2) / dist.scale**2 - 0.5, -1)
    klsum = torch.tensor([kl.double().sum().item(), float(kl.numel())], dtype=torch.float64)
    dp.sum_(klsum)  # exchange (3): KL for the learning-rate rule
    ref = torch.cat([t("grad_" + k).reshape(-1) for k, _ in model.named_parameters()])
    q.put((rank, float((flat - ref).abs().max()), float(ref.abs().max()), float(klsum[0] / klsum[1]), float(d["losses"][4]), flat.numel()))
    dp.shutdown()


def test_data_parallel_update_equals_single_process_full_batch():
    """Two gloo ranks, each with half of the environments of the reference-generated fixture, reproduce the fixture's FULL-BATCH
    parameter gradients and KL after the three all-reduces (advantage moments, flat gradient, KL sum)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale, kl, kl_ref, n in res:
        assert n == 177945
        assert err < 2e-4 * scale + 2e-6, (rank, err, scale)
        assert abs(kl - kl_ref) < 1e-3 * abs(kl_ref)


def _grid_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BG_DIST_BACKEND="gloo")
    import torch

    from booster_gym_amd.utils.parallel import DataParallel

    dp = DataParallel()
    g = torch.Generator().manual_seed(7)
    restored = torch.rand(21, 21, generator=g) * 0.6  # a non-trivial grid out of a checkpoint, the same on both ranks
    init = torch.zeros(21, 21); init[10, 10] = 1.0
    # (1) resume, no episode finished since: the grid must come back unchanged
    same = dp.sync_grid(restored.clone(), restored.clone())
    # (2) what the round-1 code did (baseline = the INITIAL grid): the restored part is counted world_size times
    wrong = dp.sync_grid(restored.clone(), init.clone())
    # (3) increments made by one rank only reach every rank once; clamped at 1
    cur = restored.clone()
    if rank == 0:
        cur[3, 4] += 0.25; cur[0, 0] += 5.0
    inc = dp.sync_grid(cur, restored.clone())
    seed = dp.broadcast_int(1234 + rank)
    q.put((rank, float((same - restored).abs().max()), float((wrong - restored).abs().max()), float(inc[3, 4] - restored[3, 4]), float(inc[0, 0]),
           float((inc - restored).abs().sum()), float(min(1.0, restored[0, 0] + 5.0) - restored[0, 0]) + 0.25, seed))
    dp.shutdown()


def test_curriculum_grid_sync_after_resume_two_ranks():
    """after a checkpoint restore the first grid sync must only carry increments made since the restore."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=180) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    for rank, same_err, wrong_err, d34, v00, total, expect_total, seed in res:
        assert same_err == 0.0
        assert wrong_err > 0.1  # the failure mode the fix removes
        assert abs(d34 - 0.25) < 1e-6 and v00 == 1.0 and abs(total - expect_total) < 1e-5
        assert seed == 1234  # rank 0's draw everywhere


def _bringup_worker(rank, world, port, q, mode):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BG_DIST_BACKEND="gloo",
                      BG_DP_LOG_ORDER="1")
    import time

    import torch
    import torch.distributed as dist

    from booster_gym_amd.utils.parallel import DataParallel, own_comm_bring_up

    dp = DataParallel()  # gloo: the process group only
    calls = []

    class FakeComm:
        def destroy(self):
            calls.append("destroy")

    def prepare(r):
        calls.append("prepare")
        if mode == "rank1_prepare_raises" and r == 1:
            raise OSError("librccl.so: cannot open shared object file (injected)")
        if mode == "rank0_id_fails" and r == 0:
            raise RuntimeError("ncclGetUniqueId failed (injected)")
        return "lib", (b"u" * 128 if r == 0 else None)

    def finish(handle, raw, r, w, dev):
        calls.append("finish")
        assert handle == "lib" and raw == b"u" * 128
        if mode == "rank1_init_raises" and r == 1:
            raise RuntimeError("ncclCommInitRank failed (injected)")
        return FakeComm()

    t0 = time.time()
    comm, err = own_comm_bring_up(rank, world, 0, "cpu", prepare, finish, timeout_s=30)
    dt = time.time() - t0
    # the process group is still in step: a collective issued by both ranks right after gives the right answer
    x = torch.tensor([float(rank + 1)])
    dist.all_reduce(x)
    # the order of the exchanges of two "mini-epochs", as the runner issues them
    for _ in range(2):
        dp.sum_(torch.zeros(3, dtype=torch.float64), tag="moments")
        dp.exchange_tail_(torch.zeros(8), torch.zeros(5, dtype=torch.float64), torch.zeros(12, dtype=torch.float64))
    q.put((rank, comm is not None, repr(err), calls, float(x.item()), dt, dp.order_log))
    dp.shutdown()


@pytest.mark.parametrize("mode", ["all_good", "rank1_prepare_raises", "rank0_id_fails", "rank1_init_raises"])
def test_own_communicator_bring_up_cannot_desynchronise_the_ranks(mode):
    """VERDICT r5 item 2 / ADVICE r5: whatever fails on one rank while the own RCCL communicator is brought up (the library missing, no unique id,
    ncclCommInitRank raising), BOTH ranks end on the same path within the timeout -- the fallback, or the communicator -- the process group's
    collectives still match afterwards, ncclCommInitRank is entered only when every rank is ready, and the ranks enqueue their exchanges in the same
    (tag, stream) order (the contract in utils/parallel.py)."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bringup_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    (r0, has0, err0, calls0, x0, dt0, log0), (r1, has1, err1, calls1, x1, dt1, log1) = res
    assert has0 == has1 == (mode == "all_good")
    assert x0 == x1 == 3.0 and dt0 < 30 and dt1 < 30
    if mode in ("rank1_prepare_raises", "rank0_id_fails"):
        assert "finish" not in calls0 and "finish" not in calls1  # nobody enters ncclCommInitRank alone
        assert "injected" in (err1 if mode == "rank1_prepare_raises" else err0)
    if mode == "rank1_init_raises":
        assert "finish" in calls0 and "destroy" in calls0 and "injected" in err1  # rank 0's communicator came up and is given back
    assert log0 == log1 and [t for t, _ in log0] == ["moments", "bucket", "moments", "bucket"]


def test_own_communicator_bring_up_is_bounded_by_a_watchdog():
    """ncclCommInitRank blocks until every rank is inside it; a rank that never arrives must not hang the others for ever: the watchdog ends the process
    with exit code 3 after BG_RCCL_INIT_TIMEOUT seconds (never a re-exec)."""
    import subprocess
    import sys

    code = (
        "import os, time, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ['PORT'])\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "from booster_gym_amd.utils.parallel import own_comm_bring_up\n"
        "own_comm_bring_up(0, 1, 0, 'cpu', lambda r: ('lib', b'u' * 128), lambda *a: time.sleep(60), timeout_s=2)\n"
        "print('returned')\n")
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PORT=str(port), PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "returned" not in p.stdout and "ncclCommInitRank has not returned" in p.stderr, (p.returncode, p.stdout, p.stderr[-400:])


def test_inline_asm_mfmas_of_the_backward_chain_keep_their_wait_states():
    """bg_mlp_chain_split_bwd.hip issues its MFMAs through inline asm (layer B's operand planes are read from accumulator registers, which the compiler's
    own MFMA will not do); the compiler does not know that those statements are MFMAs and puts operand moves directly in front of them without the
    two wait states a vector write needs before an MFMA reads the register.  The kernel pins the operand planes to their register class where they are
    assigned, so that no move is left to place; this checks the SHIPPED code object's assembly statement by statement (tools/isa_hazard_scan.py), and
    that the check itself sees the hazards when the pinning is compiled out."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_hazard_scan

    rep = isa_hazard_scan.scan()
    assert len(rep) == 2, list(rep)
    for name, (n, hazards, worst, early) in rep.items():
        assert n in (864, 2592) and not hazards and (worst is None or worst >= 2), (name, n, hazards[:3], worst)
        assert not early, (name, early[:3])  # (the counted asm loads of the next slab's input rows: untouched until the wait that makes them valid)
    bad = isa_hazard_scan.scan(["-DBG_ABL_NOPIN"])
    assert sum(len(h) for _, h, _, _ in bad.values()) > 0


def _synthetic_urdf(flat_model, path):
    """A URDF of the T1 topology written from the flat model, with the things the loader must fold: the trunk split into a root link plus two
    links behind FIXED joints (one of them rotated, one a chain of two), an inertial frame given with rpy, and a fixed sensor link on a shank."""
    m = flat_model
    L = ['<?xml version="1.0"?>', "<!-- generated by tests/test_host_logic.py -->", '<robot name="T1_synth">']

    def link(name, mass, com, i6, rpy=(0, 0, 0), shapes=()):
        L.append(f'  <link name="{name}">')
        L.append(f'    <inertial><origin xyz="{com[0]} {com[1]} {com[2]}" rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/><mass value="{mass}"/>')
        L.append(f'      <inertia ixx="{i6[0]}" iyy="{i6[1]}" izz="{i6[2]}" ixy="{i6[3]}" ixz="{i6[4]}" iyz="{i6[5]}"/></inertial>')
        for kind, size, pos in shapes:
            geo = f'<box size="{size[0]} {size[1]} {size[2]}"/>' if kind == "box" else f'<cylinder radius="{size[0]}" length="{size[1]}"/>'
            L.append(f'    <collision><origin xyz="{pos[0]} {pos[1]} {pos[2]}"/><geometry>{geo}</geometry></collision>')
        L.append("  </link>")

    def joint(name, typ, parent, child, xyz, rpy=(0, 0, 0), axis=None, lim=None):
        L.append(f'  <joint name="{name}" type="{typ}"><parent link="{parent}"/><child link="{child}"/><origin xyz="{xyz[0]} {xyz[1]} {xyz[2]}" rpy="{rpy[0]} {rpy[1]} {rpy[2]}"/>')
        if axis is not None:
            L.append(f'    <axis xyz="{axis[0]} {axis[1]} {axis[2]}"/>')
        if lim is not None:
            L.append(f'    <limit lower="{lim[0]}" upper="{lim[1]}" velocity="{lim[2]}" effort="{lim[3]}"/>')
        L.append("  </joint>")

    shp = lambda b: [(s["type"], s["size"], s["pos"]) for s in m.shapes if s["body"] == b]
    # trunk = 60 % root link + head (rotated fixed joint) + arm + hand (chain of fixed joints)
    link("Trunk", 0.6 * m.mass[0], m.com[0], 0.6 * m.inertia[0], shapes=shp(0))
    link("Head", 1.1, (0.01, 0.0, 0.05), (0.004, 0.005, 0.003, 1e-4, -2e-4, 5e-5), rpy=(0.1, -0.2, 0.3))
    joint("Head_fixed", "fixed", "Trunk", "Head", (0.02, 0.0, 0.3), rpy=(0.0, 0.3, 0.5))
    link("Arm", 1.7, (0.0, 0.05, -0.1), (0.01, 0.011, 0.002, 0, 0, 1e-4))
    joint("Arm_fixed", "fixed", "Trunk", "Arm", (0.0, 0.2, 0.2), rpy=(0.4, 0.0, 0.0))
    link("Hand", 0.4, (0.0, 0.0, -0.03), (5e-4, 5e-4, 2e-4, 0, 0, 0))
    joint("Hand_fixed", "fixed", "Arm", "Hand", (0.0, 0.02, -0.25), rpy=(0.0, -0.1, 0.2))
    ax = {1: (1, 0, 0), 2: (0, 1, 0), 3: (0, 0, 1)}
    for b in range(1, 13):
        link(m.body_names[b], m.mass[b], m.com[b], m.inertia[b], shapes=shp(b))
        j = b - 1
        joint(m.dof_names[j], "revolute", m.body_names[m.parent[b]], m.body_names[b], m.body_pos[b], axis=ax[int(m.joint_axis[b])],
              lim=(m.dof_lower[j], m.dof_upper[j], m.dof_velocity[j], m.dof_effort[j]))
        if b == 4:  # a massless-ish sensor frame on the left shank, folded into it
            link("Shank_IMU", 0.05, (0.0, 0.0, 0.0), (1e-6, 1e-6, 1e-6, 0, 0, 0))
            joint("Shank_IMU_fixed", "fixed", m.body_names[4], "Shank_IMU", (0.03, 0.0, -0.1))
    L.append("</robot>")
    open(path, "w").write("\n".join(L))


def _load_urdf_c(path, collapse=1, feet=("left_foot_link", "right_foot_link"), body_contacts=1, self_collisions=1):
    import ctypes as C

    from booster_gym_amd import _lib

    lib = _urdf_lib()
    opt = _lib.AssetOptions()
    opt.collapse_fixed_joints, opt.body_contacts, opt.self_collisions = collapse, body_contacts, self_collisions
    opt.foot_names[0], opt.foot_names[1] = feet[0].encode(), feet[1].encode()
    edge = [[0.1215, 0.05, -0.03], [0.1215, -0.05, -0.03], [-0.1015, 0.05, -0.03], [-0.1015, -0.05, -0.03]]
    for c in range(4):
        for a in range(3):
            opt.feet_edge_pos[c][a] = edge[c][a]
    h = C.c_void_p()
    rc = lib.bg_model_load_urdf(str(path).encode(), C.byref(opt), C.byref(h))
    if rc != 0:
        return rc, lib.bg_last_error().decode(), None, None
    d = _lib.ModelDesc()
    assert lib.bg_model_get(h, C.byref(d)) == 0
    names = ([lib.bg_model_body_name(h, i).decode() for i in range(d.num_bodies)], [lib.bg_model_dof_name(h, j).decode() for j in range(d.num_dofs)])
    found = (lib.bg_model_find_body(h, b"Trunk"), lib.bg_model_find_body(h, b"right_foot_link"), lib.bg_model_find_body(h, b"nope"))
    assert lib.bg_model_body_name(h, 99) is None
    lib.bg_model_destroy(h)
    return 0, d, names, found


def _compare_c_and_python_models(d, names, py):
    assert names[0] == list(py.body_names) and names[1] == list(py.dof_names)
    assert list(d.parent) == list(py.parent) and list(d.joint_axis) == list(py.joint_axis)
    f = lambda a: np.array([list(r) if hasattr(r, "__len__") else r for r in a], dtype=np.float64)
    assert np.allclose(f(d.mass), py.mass, rtol=2e-7) and np.allclose(f(d.com), py.com, atol=1e-7) and np.allclose(f(d.body_pos), py.body_pos, atol=1e-7)
    assert np.allclose(f(d.inertia), py.inertia, rtol=1e-6, atol=1e-8)
    for k in ("dof_lower", "dof_upper", "dof_velocity", "dof_effort"):
        assert np.allclose(f(getattr(d, k)), getattr(py, k), rtol=2e-7)
    sph = py.contact_spheres(exclude_bodies=(py.find_body("left_foot_link"), py.find_body("right_foot_link")))
    assert d.num_body_spheres == len(sph)
    for k, (b, c, r) in enumerate(sph):
        assert d.sphere_body[k] == b and np.allclose(list(d.sphere_pos[k]), c, atol=1e-7) and abs(d.sphere_radius[k] - r) < 1e-7
    # self-collision capsules: shank cylinders and foot boxes (t1.py:128)
    caps = py.self_collision_capsules([py.find_body("left_foot_link"), py.find_body("right_foot_link")])
    for leg in range(2):
        for k, (body, a, b, r) in enumerate(caps[leg]):
            assert np.allclose(list(d.self_capsule_a[leg][k]), a, atol=1e-7) and np.allclose(list(d.self_capsule_b[leg][k]), b, atol=1e-7)
            assert abs(d.self_capsule_r[leg][k] - r) < 1e-7 and r > 0


def test_c_urdf_loader_matches_the_python_loader(flat_model, tmp_path):
    """bg_model_load_urdf (the asset loader of the C ABI, t1.py:39-59) against booster_gym_amd.utils.urdf.load_urdf on the same file: fixed-joint
    folding with rotated frames and chains, depth-first order, limits, contact spheres; and against the packaged flat model for the legs."""
    from booster_gym_amd.utils.urdf import load_urdf

    p = tmp_path / "t1_synth.urdf"
    _synthetic_urdf(flat_model, p)
    py = load_urdf(str(p), collapse_fixed_joints=True)
    assert py.num_bodies == 13 and py.num_dofs == 12 and abs(py.mass[0] - (0.6 * flat_model.mass[0] + 1.1 + 1.7 + 0.4)) < 1e-9
    rc, d, names, found = _load_urdf_c(p)
    assert rc == 0, d
    assert found == (0, 12, -1)
    _compare_c_and_python_models(d, names, py)
    # the legs are the packaged model's (nothing folded there except the IMU frame on body 4)
    for b in (1, 2, 3, 5, 6, 7, 12):
        assert abs(d.mass[b] - flat_model.mass[b]) < 1e-6 and np.allclose(list(d.inertia[b]), flat_model.inertia[b], rtol=1e-6, atol=1e-9)
    assert abs(d.mass[4] - (flat_model.mass[4] + 0.05)) < 1e-6
    # without collapsing, the fixed joints remain: refused with the offending joint, like the Python loader
    rc, msg, _, _ = _load_urdf_c(p, collapse=0)
    assert rc == -1 and "Head_fixed" in msg and "fixed" in msg
    # malformed / unsupported inputs fail with the offending name, like the Python loader's ValueError
    bad = open(p).read().replace('<axis xyz="0 1 0"/>', '<axis xyz="0 0.7 0.7"/>', 1)
    pb = tmp_path / "bad_axis.urdf"; pb.write_text(bad)
    rc, msg, _, _ = _load_urdf_c(pb)
    assert rc == -1 and "axis" in msg
    pt = tmp_path / "truncated.urdf"; pt.write_text(open(p).read()[:2000])
    rc, msg, _, _ = _load_urdf_c(pt)
    assert rc == -1 and "XML" in msg
    rc, msg, _, _ = _load_urdf_c(tmp_path / "missing.urdf")
    assert rc == -3 and "cannot open" in msg
    # malformed joint graphs come back as errors, not as unbounded recursion: a link with two parent joints, a cycle of joints that the root
    # does not reach, a joint from a link to itself; and XML nested deeper than the reader's bound
    good = open(p).read()
    two = good.replace("</robot>", '<joint name="extra" type="fixed"><parent link="Trunk"/><child link="Hand"/></joint></robot>')
    pt2 = tmp_path / "two_parents.urdf"; pt2.write_text(two)
    rc, msg, _, _ = _load_urdf_c(pt2)
    assert rc == -1 and "more than one joint" in msg
    cyc = good.replace("</robot>", '<link name="CA"/><link name="CB"/><joint name="c1" type="fixed"><parent link="CA"/><child link="CB"/></joint>'
                       '<joint name="c2" type="fixed"><parent link="CB"/><child link="CA"/></joint></robot>')
    pc = tmp_path / "cycle.urdf"; pc.write_text(cyc)
    rc, msg, _, _ = _load_urdf_c(pc)
    assert rc == -1 and "not connected to the root" in msg
    slf = good.replace("</robot>", '<joint name="s1" type="fixed"><parent link="Hand"/><child link="Hand"/></joint></robot>')
    ps = tmp_path / "self.urdf"; ps.write_text(slf)
    rc, msg, _, _ = _load_urdf_c(ps)
    assert rc == -1 and ("itself" in msg or "more than one joint" in msg)
    pd = tmp_path / "deep.urdf"; pd.write_text("<robot>" + "<a>" * 500 + "</a>" * 500 + "</robot>")
    rc, msg, _, _ = _load_urdf_c(pd)
    assert rc == -1 and "nested deeper" in msg


def test_asset_without_leg_collision_shapes_loads_with_self_collision_off(flat_model, tmp_path):
    """An asset whose shanks carry no cylinder (no geometry to derive the leg-against-leg capsules from) is not an error: both loaders return it with
    all capsule radii 0 = "no self-collision geometry" (include/booster_gym_amd.h: bg_env_create then leaves the leg contacts off); the Python
    loader says so in a warning.  Malformed geometry that IS there stays an error (the foot box longest along y)."""
    import re

    from booster_gym_amd.utils.urdf import load_urdf

    p = tmp_path / "t1_synth.urdf"
    _synthetic_urdf(flat_model, p)
    text = open(p).read()
    bare = re.sub(r"\s*<collision>(?:(?!</collision>).)*<cylinder[^>]*/>(?:(?!</collision>).)*</collision>", "", text, flags=re.S)
    assert bare.count("<cylinder") == 0 and bare.count("<box") == text.count("<box")
    pb = tmp_path / "no_cylinders.urdf"; pb.write_text(bare)
    rc, d, names, found = _load_urdf_c(pb)
    assert rc == 0, d
    assert all(d.self_capsule_r[leg][k] == 0.0 for leg in range(2) for k in range(2))
    py = load_urdf(str(pb), collapse_fixed_joints=True)
    with pytest.warns(UserWarning, match="leg-against-leg contacts are off"):
        assert py.self_collision_capsules([py.find_body("left_foot_link"), py.find_body("right_foot_link")]) == []
    wide = text.replace('<box size="0.223 0.1 0.03"/>', '<box size="0.1 0.223 0.03"/>')
    assert wide != text
    pw = tmp_path / "wide_foot.urdf"; pw.write_text(wide)
    rc, msg, _, _ = _load_urdf_c(pw)
    assert rc == -1 and "longest along x" in msg
    with pytest.raises(ValueError, match="longest along x"):
        load_urdf(str(pw)).self_collision_capsules([6, 12])


def test_c_urdf_loader_on_the_reference_asset(flat_model):
    """The real T1 URDF (only where the reference tree is present: this container, not the GPU box): both loaders against the packaged flat model."""
    path = "/root/reference/resources/T1/T1_locomotion.urdf"
    if not os.path.isfile(path):
        pytest.skip("reference tree not present")
    from booster_gym_amd.utils.urdf import load_urdf

    py = load_urdf(path, collapse_fixed_joints=True)
    rc, d, names, found = _load_urdf_c(path)
    assert rc == 0, d
    _compare_c_and_python_models(d, names, py)
    assert names[0] == list(flat_model.body_names) and abs(sum(d.mass) - 31.6144) < 1e-3
    assert np.allclose([list(r) for r in d.inertia], flat_model.inertia, rtol=1e-5, atol=1e-8)


def test_config_keys_without_effect_are_reported():
    """Keys of the reference's T1.yaml that only configure Isaac Gym / PhysX (sim.physx.*, most of asset.*): accepted, but a value that asks
    for a behaviour this build does not have is named in a warning; the shipped values are silent.  asset.self_collisions is NOT among them
    any more: it switches the leg-against-leg contacts (envs/T1.yaml:69, envs/t1.py:128)."""
    import warnings

    from booster_gym_amd.envs.base_task import IGNORED_KEYS, warn_ignored_keys
    from booster_gym_amd.utils.config import load_cfg

    cfg = load_cfg("T1", {})
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert warn_ignored_keys(cfg) == []
    assert ("asset", "self_collisions") not in IGNORED_KEYS
    cfg2 = load_cfg("T1", {"sim.physx.num_position_iterations": 8, "asset.armature": 0.01, "sim.substeps": 2})
    with pytest.warns(UserWarning, match="no effect"):
        bad = warn_ignored_keys(cfg2)
    assert sorted(bad) == ["asset.armature", "sim.physx.num_position_iterations", "sim.substeps"]


def test_chain_split_gives_each_xcd_whole_workgroups():
    """Runner._plan_chain_split's arithmetic (host only): the two forward launches of a mini-epoch share the 256 CUs as persistent one-per-CU workgroups, and
    both counts must be multiples of the 8 XCDs -- the hardware deals workgroups round-robin over the XCDs, and 165 + 91 (the unconstrained optimum at
    every size but 4,096 envs) can hand one XCD 33 workgroups for its 32 CUs, which doubled the critic's launch (HISTORY.md round 5)."""
    from booster_gym_amd.utils.runner import plan_chain_split

    fc, fa = 64 * 256 + 256 * 256 + 256 * 128, 64 * 256 + 256 * 128 + 128 * 128  # slab costs of the critic (61 -> 256 -> 256 -> 128) and the actor (47 -> 256 -> 128 -> 128)
    for envs in (1024, 2048, 4096, 8192, 16384, 32768, 4096 + 128, 5000):
        sc, sa = (25 * envs + 127) // 128, (24 * envs + 127) // 128
        c, a = plan_chain_split(sc, sa, fc, fa, 256)
        assert c + a == 256 and c % 8 == 0 and a % 8 == 0 and c > 0 and a > 0, (envs, c, a)
        # ... at no more than 3 % above the cost of the unconstrained optimum
        cost = lambda x: max(-(-sc // x) * fc, -(-sa // (256 - x)) * fa)
        assert cost(c) <= 1.03 * min(cost(x) for x in range(1, 256)), (envs, c, cost(c))
    assert plan_chain_split(800, 768, fc, fa, 256) == (160, 96)  # the bench shape
    assert sum(plan_chain_split(800, 768, fc, fa, 250)) == 250   # a part whose CU count is no multiple of 8: any split
