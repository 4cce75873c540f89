import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues per process (default 4).  With the process group's own streams beside this build's
# two (the critic's work runs on a high-priority side stream, utils/runner.py) the default mapping costs the update phase 50 % of its speed:
# measured on one MI355X with RCCL in a world of one rank, 32-33 ms per update against 21.4 ms without a process group (29.7 ms even with every
# collective skipped), and 22.4 ms with 8 (or 2) hardware queues, collectives included (tools/host_enqueue_probe2.py).
# The variable is read when the HIP runtime starts, so it is set here, when the package is imported, for processes that will join a process
# group (the launcher's WORLD_SIZE, or BG_DIST_FORCE); a value from the environment wins.  Without a process group it changes nothing (measured).
HW_QUEUES_SET_TOO_LATE = False  # the setting below was made after this process had already started the HIP runtime (utils/parallel.py warns)
if int(_os.environ.get("WORLD_SIZE", "1")) > 1 or _os.environ.get("BG_DIST_FORCE", "0") == "1":
    if "GPU_MAX_HW_QUEUES" not in _os.environ:
        import torch as _torch

        HW_QUEUES_SET_TOO_LATE = bool(_torch.cuda.is_initialized())
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"
