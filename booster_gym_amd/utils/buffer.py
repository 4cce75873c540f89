"""Rollout storage: a dict of `[T(+extra), N, *shape]` device tensors (reference utils/buffer.py:4-25).

`extra_rows=1` keeps one more time row so that the observation following the last step of a rollout lives in
the same tensor (row T); the HIP env kernel writes step outputs straight into these rows.
"""
import torch


class ExperienceBuffer:
    def __init__(self, horizon_length, num_envs, device):
        self.tensor_dict = {}
        self.horizon_length = horizon_length
        self.num_envs = num_envs
        self.device = device

    def add_buffer(self, name, shape, dtype=None, extra_rows=0):
        self.tensor_dict[name] = torch.zeros(self.horizon_length + extra_rows, self.num_envs, *shape, dtype=dtype, device=self.device)

    def update_data(self, name, idx, data):
        self.tensor_dict[name][idx, :] = data

    def __len__(self):
        return len(self.tensor_dict)

    def __getitem__(self, buf_name):
        return self.tensor_dict[buf_name]

    def keys(self):
        return self.tensor_dict.keys()
