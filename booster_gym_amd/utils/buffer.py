"""Rollout storage for the fused rollout (fills the role of reference utils/buffer.py:4-25; `ExperienceBuffer`, `add_buffer` and
`buf[name]` are the names utils/runner.py:36-42,112-121 uses).

What this build needs from it is different from the reference's dict of tensors that the runner copies into step by step:
  * every stream is allocated ONCE as `[T + carry][N][*shape]` device memory and never re-bound: the HIP env kernel and the actor
    kernel receive raw row pointers (`row(name, t)`) and write step outputs in place;
  * streams registered with `carry=True` (observations) own one extra time row: row T holds the observation after the last step,
    which is both the critic's bootstrap input and, after `roll()`, row 0 of the next rollout;
  * `flat(name)` is the `[T*N][*shape]` view the full-batch update kernels read (no copy; rows [T*N, (T+1)*N) of a carried
    stream are reachable through `flat(name, with_carry=True)`).
"""
import torch


class ExperienceBuffer:
    def __init__(self, horizon_length, num_envs, device):
        self.horizon_length, self.num_envs, self.device = int(horizon_length), int(num_envs), device
        self._streams = {}
        self._carried = []

    def add_buffer(self, name, shape, dtype=None, extra_rows=0, carry=None):
        carry = bool(extra_rows) if carry is None else carry
        if name in self._streams:
            raise KeyError(f"stream {name!r} already registered")
        t = torch.zeros(self.horizon_length + (1 if carry else 0), self.num_envs, *shape, dtype=dtype, device=self.device)
        self._streams[name] = t
        if carry:
            self._carried.append(name)
        return t

    def __getitem__(self, name):
        return self._streams[name]

    def __contains__(self, name):
        return name in self._streams

    def names(self):
        return tuple(self._streams)

    def row(self, name, t):
        """Time row t of a stream ([N][*shape], contiguous): what a kernel launch writes into."""
        return self._streams[name][t]

    def flat(self, name, with_carry=False):
        s = self._streams[name]
        rows = s.shape[0] if with_carry else self.horizon_length
        return s[:rows].reshape(rows * self.num_envs, *s.shape[2:])

    def roll(self):
        """End of an iteration: the carried row T becomes row 0 of the next rollout."""
        T = self.horizon_length
        for name in self._carried:
            s = self._streams[name]
            s[0].copy_(s[T])

    def nbytes(self):
        return sum(s.numel() * s.element_size() for s in self._streams.values())
