"""The Isaac Gym tensor-API calls of the reference's envs/t1.py, one method per call, on the HIP simulator.

This is the lower seam of the drop-in boundary: a maintainer who keeps the reference's Python task logic
replaces `self.gym.<call>(self.sim, ...)` by `self.gym.<call>(...)` on a `GymCalls` object and `gymtorch.wrap_tensor(...)` by the
tensors `acquire_*` returns.  Method names and argument meaning follow Isaac Gym as used at the cited lines; the fused `T1.step`
runs the same physics (and the whole task logic) in one launch.
"""
import torch

from .. import _lib


class GymCalls:
    def __init__(self, env):
        """env: a booster_gym_amd.envs.T1 (owns the model, per-env parameters and terrain)."""
        self._owner = env
        self._lib = env._lib
        self._env = env._env
        n, dev = env.num_envs, env.device
        self.num_envs = n
        self._root = torch.zeros(n, 13, dtype=torch.float32, device=dev)
        self._root[:, 6] = 1.0
        self._dof = torch.zeros(n, _lib.NUM_DOFS, 2, dtype=torch.float32, device=dev)
        self._contact = torch.zeros(n, _lib.NUM_BODIES, 3, dtype=torch.float32, device=dev)
        self._body = torch.zeros(n, _lib.NUM_BODIES, 13, dtype=torch.float32, device=dev)
        _lib.check(self._lib.bg_sim_bind_state(self._env, _lib.ptr(self._root), _lib.ptr(self._dof), _lib.ptr(self._contact), _lib.ptr(self._body)),
                   "bg_sim_bind_state")

    # ---- acquire_*: the tensors ARE the simulator state (t1.py:203-220)
    def acquire_actor_root_state_tensor(self):
        return self._root

    def acquire_dof_state_tensor(self):
        return self._dof.view(self.num_envs * _lib.NUM_DOFS, 2)

    def acquire_net_contact_force_tensor(self):
        return self._contact.view(self.num_envs * _lib.NUM_BODIES, 3)

    def acquire_rigid_body_state_tensor(self):
        return self._body.view(self.num_envs * _lib.NUM_BODIES, 13)

    # ---- t1.py:450-451
    def set_dof_actuation_force_tensor(self, torques):
        t = self._f32(torques, self.num_envs * _lib.NUM_DOFS)
        _lib.check(self._lib.bg_sim_set_actuation(self._env, _lib.ptr(t), _lib.current_stream_ptr()), "bg_sim_set_actuation")

    def simulate(self):
        _lib.check(self._lib.bg_sim_simulate(self._env, _lib.current_stream_ptr()), "bg_sim_simulate")

    def fetch_results(self, wait=True):  # t1.py:452-453: work is ordered on the stream
        pass

    # ---- t1.py:454-455, 460-462: the bound tensors are written by simulate()
    def refresh_dof_state_tensor(self):
        pass

    def refresh_actor_root_state_tensor(self):
        pass

    def refresh_net_contact_force_tensor(self):
        pass

    def refresh_rigid_body_state_tensor(self):
        _lib.check(self._lib.bg_sim_refresh_body_state(self._env, _lib.current_stream_ptr()), "bg_sim_refresh_body_state")

    # ---- t1.py:522-527 (space must be LOCAL_SPACE, the only one the reference uses)
    def apply_rigid_body_force_tensors(self, forces=None, torques=None, space="LOCAL_SPACE"):
        if space not in ("LOCAL_SPACE", 1):
            raise ValueError("apply_rigid_body_force_tensors: only LOCAL_SPACE is supported")
        k = self.num_envs * _lib.NUM_BODIES * 3
        f = None if forces is None else self._f32(forces, k)
        t = None if torques is None else self._f32(torques, k)
        _lib.check(self._lib.bg_sim_apply_body_wrench_local(self._env, _lib.ptr(f), _lib.ptr(t), _lib.current_stream_ptr()),
                   "bg_sim_apply_body_wrench_local")

    # ---- t1.py:341,359,504 and t1.py:323-325
    def set_actor_root_state_tensor_indexed(self, root_states, env_ids_int32, count=None):
        self._write_back(root_states, self._root, env_ids_int32, count, self._lib.bg_sim_write_root_state, "bg_sim_write_root_state")

    def set_dof_state_tensor_indexed(self, dof_state, env_ids_int32, count=None):
        self._write_back(dof_state, self._dof, env_ids_int32, count, self._lib.bg_sim_write_dof_state, "bg_sim_write_dof_state")

    def set_actor_root_state_tensor(self, root_states):
        ids = torch.arange(self.num_envs, dtype=torch.int32, device=self._root.device)
        self.set_actor_root_state_tensor_indexed(root_states, ids)

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
