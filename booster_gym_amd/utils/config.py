"""Config loading: one YAML per task, CLI overrides on top (reference utils/runner.py:44-68)."""
import copy
import os

import yaml

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def config_path(task):
    """`envs/<task>.yaml` relative to the working directory (reference behaviour), else the packaged file."""
    local = os.path.join("envs", f"{task}.yaml")
    if os.path.isfile(local):
        return local
    pk = os.path.join(_PKG, "envs", f"{task}.yaml")
    if os.path.isfile(pk):
        return pk
    raise FileNotFoundError(f"no config for task {task!r}: tried {local} and {pk}")


def load_cfg(task="T1", overrides=None):
    with open(config_path(task), "r", encoding="utf-8") as f:
        cfg = yaml.load(f.read(), Loader=yaml.FullLoader)
    cfg["basic"]["task"] = task
    for dotted, value in (overrides or {}).items():
        node = cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        node[keys[-1]] = copy.deepcopy(value)
    return cfg
