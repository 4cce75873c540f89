"""BaseTask: device routing and terrain creation.

Mirrors reference `envs/base_task.py:7-79` (`BaseTask.__init__`, `create_sim`).  The viewer / camera half of
the reference class (`set_viewer`, `render`, base_task.py:81-140) has no counterpart: a headless MI355X
node has no GL context and training forces `record_video=False` (reference runner.py:67-68).
"""
from ..utils.terrain import Terrain


class BaseTask:
    def __init__(self, cfg):
        self.cfg = cfg
        self.create_sim()
        self.terrain = Terrain(self.device, self.cfg["terrain"], seed=int(self.cfg["basic"].get("seed", 0)))
        self.viewer = None
        self.camera = None
        self.camera_frames = []

    def create_sim(self):
        sim_cfg = self.cfg["sim"]
        sim_device = str(self.cfg["basic"]["sim_device"])
        dev_type, _, idx = sim_device.partition(":")
        if dev_type != "cuda":
            # reference base_task.py:26-29 falls back to PhysX-CPU; this build has only the HIP simulator
            raise ValueError(
                f"sim_device={sim_device!r}: booster_gym_amd simulates on an AMD GPU only (use 'cuda:<n>'); "
                "the CPU restatement under oracle/ is test infrastructure, not a product path"
            )
        self.sim_device_id = int(idx) if idx else 0
        self.device = f"cuda:{self.sim_device_id}"
        self.headless = self.cfg["basic"].get("headless", True)
        if sim_cfg["up_axis"] == "z":
            self.up_axis_idx = 2
        else:
            raise ValueError(f"Invalid physics up-axis: {sim_cfg['up_axis']} (the T1 model and its rewards assume z-up)")
        if sim_cfg.get("physics_engine", "hip_aba") not in ("hip_aba", "physx", "flex"):
            raise ValueError(f"Invalid physics engine backend: {sim_cfg['physics_engine']}")
        self.sim_params = {"dt": sim_cfg["dt"], "gravity": list(sim_cfg["gravity"]), "substeps": sim_cfg.get("substeps", 1)}

    def render(self):
        return None
