"""Throw-away stand-in for the absent third-party `isaacgym` package, used ONLY by tests/golden/make_task_fixtures.py in the
build container so that the reference's task code (envs/t1.py, utils/terrain.py) can be imported and its pure-torch
methods called on hand-made inputs.  `gymapi` / `gymtorch` / `gymutil` / `terrain_utils` are empty; `torch_utils` restates the
handful of math helpers the task code imports (semantics from general knowledge of Isaac Gym Preview 4,
only `quat_rotate_inverse` is pinned by the reference itself, play_mujoco.py:282-297).  Never imported by the product."""
import types

gymapi = types.ModuleType("isaacgym.gymapi")
gymtorch = types.ModuleType("isaacgym.gymtorch")
gymutil = types.ModuleType("isaacgym.gymutil")
terrain_utils = types.ModuleType("isaacgym.terrain_utils")
