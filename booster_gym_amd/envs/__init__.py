"""Task registry: the runner resolves `cfg["basic"]["task"]` by name here (reference envs/__init__.py:1, runner.py:27)."""
from .t1 import T1

TASKS = {"T1": T1}
