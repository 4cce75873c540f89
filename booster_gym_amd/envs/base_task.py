"""BaseTask: device routing and terrain creation.

Mirrors reference `envs/base_task.py:7-79` (`BaseTask.__init__`, `create_sim`).  The viewer / camera half of
the reference class (`set_viewer`, `render`, base_task.py:81-140) has no counterpart: a headless MI355X
node has no GL context and training forces `record_video=False` (reference runner.py:67-68).
"""
import warnings

from ..utils.terrain import Terrain

# Keys of the reference's envs/T1.yaml that configure Isaac Gym / PhysX and have NO effect on this build's simulator, with the value that
# claims nothing (the shipped one).  A different value asks for a behaviour that will not happen: say so once instead of accepting it silently.
IGNORED_KEYS = {
    ("sim", "substeps"): 1,
    ("sim", "physx", "solver_type"): 1, ("sim", "physx", "num_position_iterations"): 4, ("sim", "physx", "num_velocity_iterations"): 1,
    ("sim", "physx", "contact_offset"): 0.02, ("sim", "physx", "rest_offset"): 0.0, ("sim", "physx", "bounce_threshold_velocity"): 0.2,
    ("sim", "physx", "max_depenetration_velocity"): 100.0, ("sim", "physx", "contact_collection"): 1,
    ("asset", "replace_cylinder_with_capsule"): False, ("asset", "flip_visual_attachments"): False, ("asset", "fix_base_link"): False,
    ("asset", "disable_gravity"): False, ("asset", "default_dof_drive_mode"): 3, ("asset", "angular_damping"): 0.0, ("asset", "linear_damping"): 0.0,
    ("asset", "armature"): 0.0, ("asset", "thickness"): 0.01, ("asset", "density"): 0.001, ("asset", "max_angular_velocity"): 1000.0,
    ("asset", "max_linear_velocity"): 1000.0,
}


def warn_ignored_keys(cfg):
    """One warning naming every key of IGNORED_KEYS whose value differs from the neutral one (returns the list of offending key paths)."""
    bad = []
    for path, neutral in IGNORED_KEYS.items():
        node = cfg
        for k in path:
            node = node.get(k, None) if isinstance(node, dict) else None
            if node is None:
                break
        if node is not None and node != neutral:
            bad.append((".".join(path), node, neutral))
    if bad:
        warnings.warn("these config keys configure Isaac Gym / PhysX and have no effect on the HIP simulator of booster_gym_amd: " +
                      ", ".join(f"{k} = {v!r} (only {n!r} claims nothing)" for k, v, n in bad), stacklevel=3)
    return [k for k, _, _ in bad]


class BaseTask:
    def __init__(self, cfg):
        self.cfg = cfg
        self.ignored_keys = warn_ignored_keys(cfg)
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
