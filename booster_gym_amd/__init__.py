import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues per process (default 4).  With a process group's streams beside this build's two (the
# critic's work runs on a high-priority side stream, utils/runner.py) the default mapping costs the whole loop 50-65 % of its speed.  Measured on one
# MI355X with RCCL in a world of one rank (tools/ab_env.sh): round 3, every exchange through torch.distributed's NCCL backend:
# 41 ms per iteration at 4 queues, 25.4 ms at 8 or 2; round 5, the per-mini-epoch exchanges through an own communicator on the update's streams
# (utils/rccl.py): 39.8 ms at 8, 12 or 16 queues (the env step itself runs at 164 us instead of 101), 24.5-24.6 ms at 1, 2 or 3 -- against 23.9-24.0 ms
# without a process group.  2 is the value that was good in every configuration measured.
# The variable is read when the HIP runtime starts, so it is set here, when the package is imported, for processes that will join a process
# group (the launcher's WORLD_SIZE, or BG_DIST_FORCE); a value from the environment wins.  Without a process group it changes nothing (measured).
HW_QUEUES_SET_TOO_LATE = False  # the setting below was made after this process had already started the HIP runtime (utils/parallel.py warns)
if int(_os.environ.get("WORLD_SIZE", "1")) > 1 or _os.environ.get("BG_DIST_FORCE", "0") == "1":
    if "GPU_MAX_HW_QUEUES" not in _os.environ:
        import torch as _torch

        HW_QUEUES_SET_TOO_LATE = bool(_torch.cuda.is_initialized())
        _os.environ["GPU_MAX_HW_QUEUES"] = "2"
