"""The Isaac Gym tensor-API calls of the reference's envs/t1.py, one method per call, on the HIP simulator.

This is the lower seam of the drop-in boundary.  Every method takes the arguments the reference passes at the cited
line -- the leading `sim` handle, `gymtorch.unwrap_tensor(...)` wrappers, `gymapi.LOCAL_SPACE` -- so that a maintainer who keeps the reference's
Python task logic only rebinds three names:

    from booster_gym_amd.envs.gym_calls import GymCalls, gymtorch, gymapi     # instead of: from isaacgym import gymtorch, gymapi
    self.gym = GymCalls(env); self.sim = self.gym.sim                         # instead of: acquire_gym() / create_sim()

and leaves lines such as `self.gym.set_dof_actuation_force_tensor(self.sim, gymtorch.unwrap_tensor(dof_torques))` (t1.py:450) untouched.
The fused `T1.step` runs the same physics (and the whole task logic) in one launch.
"""
import enum

import numpy as np
import torch

from .. import _lib


class CoordinateSpace(enum.IntEnum):
    """gymapi.CoordinateSpace as the reference uses it (t1.py:526: LOCAL_SPACE)."""
    ENV_SPACE = 0
    LOCAL_SPACE = 1
    GLOBAL_SPACE = 2


class _GymApi:
    ENV_SPACE, LOCAL_SPACE, GLOBAL_SPACE = CoordinateSpace.ENV_SPACE, CoordinateSpace.LOCAL_SPACE, CoordinateSpace.GLOBAL_SPACE


class _GymTorch:
    """gymtorch.wrap_tensor / unwrap_tensor (t1.py:215-220, 323-325, 450): the simulator's tensors ARE torch tensors here, so both are the identity."""

    @staticmethod
    def wrap_tensor(t):
        return t

    @staticmethod
    def unwrap_tensor(t):
        return t


gymapi = _GymApi()
gymtorch = _GymTorch()


class SimHandle:
    """What `self.sim` holds in the reference: opaque, passed back as the first argument of every call."""

    def __init__(self, owner):
        self.owner = owner


class GymCalls:
    def __init__(self, env):
        """env: a booster_gym_amd.envs.T1 (owns the model, per-env parameters and terrain)."""
        self._owner = env
        self._lib = env._lib
        self._env = env._env
        self.sim = SimHandle(self)
        n, dev = env.num_envs, env.device
        self.num_envs = n
        self._root = torch.zeros(n, 13, dtype=torch.float32, device=dev)
        self._root[:, 6] = 1.0
        self._dof = torch.zeros(n, _lib.NUM_DOFS, 2, dtype=torch.float32, device=dev)
        self._contact = torch.zeros(n, _lib.NUM_BODIES, 3, dtype=torch.float32, device=dev)
        self._body = torch.zeros(n, _lib.NUM_BODIES, 13, dtype=torch.float32, device=dev)
        _lib.check(self._lib.bg_sim_bind_state(self._env, _lib.ptr(self._root), _lib.ptr(self._dof), _lib.ptr(self._contact), _lib.ptr(self._body)),
                   "bg_sim_bind_state")

    def _check_sim(self, sim):
        if sim is not self.sim:
            raise TypeError("first argument must be this simulator's `sim` handle (GymCalls.sim), as in self.gym.<call>(self.sim, ...)")

    # ---- asset queries (t1.py:54-59, 85-108): the asset is the env's flat model (utils/urdf.py / bg_model_load_urdf)
    def load_asset(self, sim, asset_root=None, asset_file=None, asset_options=None):
        self._check_sim(sim)
        return self._owner.model

    def get_asset_dof_count(self, asset):
        return asset.num_dofs

    def get_asset_rigid_body_count(self, asset):
        return asset.num_bodies

    def get_asset_dof_names(self, asset):
        return list(asset.dof_names)

    def get_asset_rigid_body_names(self, asset):
        return list(asset.body_names)

    def find_asset_rigid_body_index(self, asset, name):
        return asset.find_body(name)

    def get_asset_dof_properties(self, asset):
        """Structured array with the fields t1.py:59-67 reads (`lower`, `upper`, `velocity`, `effort`) plus the ones it writes."""
        p = np.zeros(asset.num_dofs, dtype=[(k, np.float32) for k in ("lower", "upper", "velocity", "effort", "stiffness", "damping", "friction", "armature")])
        p["lower"], p["upper"], p["velocity"], p["effort"] = asset.dof_lower, asset.dof_upper, asset.dof_velocity, asset.dof_effort
        return p

    def prepare_sim(self, sim):  # t1.py:29
        self._check_sim(sim)

    # ---- acquire_*: the tensors ARE the simulator state (t1.py:203-220)
    def acquire_actor_root_state_tensor(self, sim):
        self._check_sim(sim)
        return self._root

    def acquire_dof_state_tensor(self, sim):
        self._check_sim(sim)
        return self._dof.view(self.num_envs * _lib.NUM_DOFS, 2)

    def acquire_net_contact_force_tensor(self, sim):
        self._check_sim(sim)
        return self._contact.view(self.num_envs * _lib.NUM_BODIES, 3)

    def acquire_rigid_body_state_tensor(self, sim):
        self._check_sim(sim)
        return self._body.view(self.num_envs * _lib.NUM_BODIES, 13)

    # ---- t1.py:450-451
    def set_dof_actuation_force_tensor(self, sim, torques):
        self._check_sim(sim)
        t = self._f32(torques, self.num_envs * _lib.NUM_DOFS)
        _lib.check(self._lib.bg_sim_set_actuation(self._env, _lib.ptr(t), _lib.current_stream_ptr()), "bg_sim_set_actuation")

    def simulate(self, sim):
        self._check_sim(sim)
        _lib.check(self._lib.bg_sim_simulate(self._env, _lib.current_stream_ptr()), "bg_sim_simulate")

    def fetch_results(self, sim, wait=True):  # t1.py:452-453: work is ordered on the stream
        self._check_sim(sim)

    # ---- t1.py:454-455, 460-462: the bound tensors are written by simulate()
    def refresh_dof_state_tensor(self, sim):
        self._check_sim(sim)

    def refresh_dof_force_tensor(self, sim):  # t1.py:211,455: force sensors are enabled but never read
        self._check_sim(sim)

    def refresh_actor_root_state_tensor(self, sim):
        self._check_sim(sim)

    def refresh_net_contact_force_tensor(self, sim):
        self._check_sim(sim)

    def refresh_rigid_body_state_tensor(self, sim):
        self._check_sim(sim)
        _lib.check(self._lib.bg_sim_refresh_body_state(self._env, _lib.current_stream_ptr()), "bg_sim_refresh_body_state")

    # ---- t1.py:522-527 (space must be LOCAL_SPACE, the only one the reference uses)
    def apply_rigid_body_force_tensors(self, sim, forces=None, torques=None, space=CoordinateSpace.LOCAL_SPACE):
        self._check_sim(sim)
        if space not in ("LOCAL_SPACE", CoordinateSpace.LOCAL_SPACE):
            raise ValueError("apply_rigid_body_force_tensors: only gymapi.LOCAL_SPACE is supported")
        k = self.num_envs * _lib.NUM_BODIES * 3
        f = None if forces is None else self._f32(forces, k)
        t = None if torques is None else self._f32(torques, k)
        _lib.check(self._lib.bg_sim_apply_body_wrench_local(self._env, _lib.ptr(f), _lib.ptr(t), _lib.current_stream_ptr()),
                   "bg_sim_apply_body_wrench_local")

    # ---- t1.py:341,359,504 and t1.py:323-325
    def set_actor_root_state_tensor_indexed(self, sim, root_states, env_ids_int32, count=None):
        self._check_sim(sim)
        self._write_back(root_states, self._root, env_ids_int32, count, self._lib.bg_sim_write_root_state, "bg_sim_write_root_state")

    def set_dof_state_tensor_indexed(self, sim, dof_state, env_ids_int32, count=None):
        self._check_sim(sim)
        self._write_back(dof_state, self._dof, env_ids_int32, count, self._lib.bg_sim_write_dof_state, "bg_sim_write_dof_state")

    def set_actor_root_state_tensor(self, sim, root_states):
        ids = torch.arange(self.num_envs, dtype=torch.int32, device=self._root.device)
        self.set_actor_root_state_tensor_indexed(sim, root_states, ids)

    def _write_back(self, src, bound, ids, count, fn, what):
        ids = ids.to(dtype=torch.int32, device=bound.device).contiguous()
        count = int(ids.numel() if count is None else count)
        if src.data_ptr() != bound.data_ptr():  # a caller-side copy of the state tensor: take the listed rows
            rows = ids[:count].long()
            bound.view(self.num_envs, -1)[rows] = src.reshape(self.num_envs, -1).to(bound.dtype)[rows]
        _lib.check(fn(self._env, _lib.ptr(ids), count, _lib.current_stream_ptr()), what)

    def _f32(self, t, numel):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self._root.device:
            t = t.to(device=self._root.device, dtype=torch.float32).contiguous()
        if t.numel() != numel:
            raise ValueError(f"expected {numel} elements, got {tuple(t.shape)}")
        return t
