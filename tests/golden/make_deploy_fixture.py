"""Numbers-only fixture of the deployed policy's I/O contract: tests/golden/deploy_contract.json.

Source: /root/reference/deploy/configs/T1.yaml (the configuration deploy/utils/policy.py:34-73 feeds the exported actor with on the robot): control step,
decimation, action scale and clip, observation scales, and the leg entries (joints 11..22 of the 23-joint robot order = the 12 policy DoFs,
deploy/utils/policy.py:60) of the PD gains, default pose and torque limits.  Only numbers and key names are stored.

Run in the build container:  python tests/golden/make_deploy_fixture.py
"""
import json

import yaml

SRC = "/root/reference/deploy/configs/T1.yaml"
d = yaml.safe_load(open(SRC))
c, p = d["common"], d["policy"]
legs = slice(11, 23)
out = {
    "dt": c["dt"], "decimation": p["control"]["decimation"], "action_scale": p["control"]["action_scale"],
    "num_actions": p["num_actions"], "num_observations": p["num_observations"], "gait_frequency": p["gait_frequency"],
    "normalization": {k: p["normalization"][k] for k in ("gravity", "lin_vel", "ang_vel", "dof_pos", "dof_vel", "clip_actions")},
    "leg_stiffness": c["stiffness"][legs], "leg_damping": c["damping"][legs], "leg_default_qpos": c["default_qpos"][legs],
    "leg_torque_limit": c["torque_limit"][legs],
}
json.dump(out, open("tests/golden/deploy_contract.json", "w"), indent=1)
print(out)
