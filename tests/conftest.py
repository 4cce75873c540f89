import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def flat_model():
    from booster_gym_amd.utils.urdf import FlatModel

    return FlatModel.load(os.path.join(ROOT, "booster_gym_amd", "resources", "T1", "T1_locomotion.flat.json"))


# left / right mirror map of the T1 legs: pitch joints (about y) keep their sign under y -> -y, roll (x) and yaw (z) joints change it
MIRROR_SIGN = (1.0, -1.0, -1.0, 1.0, 1.0, -1.0)


def symmetrised(model):
    """A copy of the flat model whose right leg is the exact mirror image of its left leg and whose trunk is symmetric about the sagittal plane.
    The shipped inertials are not (shank 1.73 / 1.79 kg, reference resources/T1/T1_locomotion.xml:63 vs 96, 68 vs 101; trunk com 8 um off the plane);
    the collision primitives and the sole corners already are.  For the mirrored-pose known-answer tests."""
    import copy

    import numpy as np

    sym = copy.deepcopy(model)
    my, mi = np.array([1.0, -1.0, 1.0]), np.array([1.0, 1.0, 1.0, -1.0, 1.0, -1.0])  # y -> -y on vectors; on (xx, yy, zz, xy, xz, yz)
    for i in range(6):
        a, b = 1 + i, 7 + i
        sym.mass[b], sym.com[b], sym.body_pos[b], sym.inertia[b] = sym.mass[a], sym.com[a] * my, sym.body_pos[a] * my, sym.inertia[a] * mi
    sym.com[0, 1] = 0.0
    sym.inertia[0, 3] = 0.0
    sym.inertia[0, 5] = 0.0
    return sym


def mirrored_states(rng, n, standing, z_range=(0.64, 0.70)):
    """n states that are their own mirror image: trunk pitched only, velocities in the sagittal plane, right-leg joints / rates / torques = MIRROR_SIGN x
    the left leg's.  standing: around the default pose with the soles on (z_range: how deep in) the plane z = 0; else airborne (the legs may touch
    each other)."""
    import numpy as np

    S = np.array(MIRROR_SIGN)
    root, q, qd, tau = np.zeros((n, 13)), np.zeros((n, 12)), np.zeros((n, 12)), np.zeros((n, 12))
    for e in range(n):
        if standing:
            qL = np.array([-0.2, 0, 0, 0.4, -0.25, 0]) + rng.normal(size=6) * 0.05
            qdL, tL, z, vs = rng.normal(size=6) * 0.3, rng.uniform(-10, 10, 6), rng.uniform(*z_range), 0.2
        else:
            qL, qdL, tL, z, vs = rng.uniform(-0.4, 0.4, 6), rng.normal(size=6), rng.uniform(-20, 20, 6), 5.0, 1.0
        q[e], qd[e], tau[e] = np.concatenate([qL, S * qL]), np.concatenate([qdL, S * qdL]), np.concatenate([tL, S * tL])
        th = rng.uniform(-0.3, 0.3) * (0.2 if standing else 1.0)
        root[e, 2] = z
        root[e, 3:7] = [0.0, np.sin(th / 2), 0.0, np.cos(th / 2)]
        root[e, 7:10] = [rng.normal() * vs, 0.0, rng.normal() * vs]
        root[e, 10:13] = [0.0, rng.normal() * vs, 0.0]
    return root, q, qd, tau
