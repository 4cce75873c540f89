"""Produce the packaged flat model (numbers only) from a URDF.

Usage: python tools/flatten_urdf.py /root/reference/resources/T1/T1_locomotion.urdf \
           booster_gym_amd/resources/T1/T1_locomotion.flat.json
The URDF itself is not copied into this repository; only the collapsed
13-body numeric model produced by booster_gym_amd.utils.urdf.load_urdf is.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from booster_gym_amd.utils.urdf import load_urdf

if __name__ == "__main__":
    m = load_urdf(sys.argv[1], collapse_fixed_joints=True)
    m.save(sys.argv[2])
    print(f"{m.name}: {m.num_bodies} bodies, {m.num_dofs} dofs, total mass {m.mass.sum():.6f} kg")
