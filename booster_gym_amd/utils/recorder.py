"""Run directory, scalar logging and checkpoints (reference utils/recorder.py:9-79).

Layout and names are the reference's: `logs/<timestamp>/{config.yaml, nn/model_<it>.pth, summaries/}`; scalars `steps`,
`reward`, `episode/<term>`, and whatever the runner passes to `record_statistics`.  TensorBoard / W&B are optional
dependencies that are absent from the MI355X image: scalars always go to `summaries/scalars.jsonl`, and to a
`SummaryWriter` as well when `torch.utils.tensorboard` imports.  Episode statistics are accumulated on the device by
the env kernel (no per-done-env `.item()` loop, reference recorder.py:41-53) and read once per iteration.
"""
import json
import os
import time

import torch
import yaml


class Recorder:
    def __init__(self, cfg, root="logs", rank=0):
        self.cfg = cfg
        self.rank = rank
        self.enabled = rank == 0
        self.writer = None
        self._jsonl = None
        if not self.enabled:
            return
        name = time.strftime("%Y-%m-%d-%H-%M-%S", time.localtime())
        self.dir = os.path.join(root, name)
        os.makedirs(self.dir, exist_ok=True)
        self.model_dir = os.path.join(self.dir, "nn")
        os.makedirs(self.model_dir, exist_ok=True)
        sdir = os.path.join(self.dir, "summaries")
        os.makedirs(sdir, exist_ok=True)
        self._jsonl = open(os.path.join(sdir, "scalars.jsonl"), "a", buffering=1)
        try:
            from torch.utils.tensorboard import SummaryWriter  # optional

            self.writer = SummaryWriter(sdir)
        except Exception:
            self.writer = None
        with open(os.path.join(self.dir, "config.yaml"), "w") as file:
            yaml.dump(self.cfg, file)

    def _scalar(self, path, value, it):
        if not self.enabled:
            return
        self._jsonl.write(json.dumps({"tag": path, "value": float(value), "step": int(it)}) + "\n")
        if self.writer is not None:
            self.writer.add_scalar(path, float(value), it)

    def record_episode_statistics(self, env, reward_names, it, stats=None):
        """Flush the device-side episode accumulators: mean over the episodes that ended since the last call.  `stats`: the accumulator
        values already on the host (the runner reads them without stalling, one iteration late); default: read them now."""
        s = env.episode_stats(reset=True).cpu().tolist() if stats is None else list(stats)
        n = s[0]
        mean = (lambda v: v / n) if n > 0 else (lambda v: 0.0)
        out = {"steps": mean(s[1]), "reward": mean(s[2])}
        from .. import _lib

        for name in reward_names:
            out["episode/" + name] = mean(s[3 + _lib.REWARD_NAMES.index(name)])
        out["episodes_finished"] = n
        out["nonfinite_resets"] = s[-1]
        for k, v in out.items():
            self._scalar(k, v, it)
        return out

    def record_statistics(self, statistics, it):
        for key, value in statistics.items():
            self._scalar(key, float(value), it)

    def save(self, model_dict, it):
        if not self.enabled:
            return None
        path = os.path.join(self.model_dir, "model_{}.pth".format(it))
        print("Saving model to {}".format(path))
        torch.save(model_dict, path)
        return path
