"""Numbers-only fixture of the reference's MuJoCo model of the T1: tests/golden/mjcf_model.npz.

Source: /root/reference/resources/T1/T1_locomotion.xml:36-139 (the model `play_mujoco.py:717-720` steps).  The reference holds no
dynamics vectors; this file and the trained actor are the two artefacts in its tree that embed its physics (KAT 1).
Stored per body in document order (= depth-first, the DoF order of T1_locomotion.xml:123-134): parent index, `pos`, mass, inertial `pos`,
the full inertia tensor about the centre of mass in body axes (R diag(diaginertia) R^T with R from the inertial `quat`, w x y z); per hinge:
axis and range; per motor: ctrlrange; every non-mesh geom (the collision primitives): body, type code, MuJoCo half-sizes, pos.
Only numbers and the body / joint names are stored, no XML text.

Run in the build container:  python tests/golden/make_model_fixture.py
"""
import xml.etree.ElementTree as ET

import numpy as np

SRC = "/root/reference/resources/T1/T1_locomotion.xml"
GEOM_CODE = {"plane": 0, "box": 1, "cylinder": 2, "sphere": 3, "capsule": 4}


def quat_to_mat(q):
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def vec(s, n):
    v = np.array([float(t) for t in s.split()], dtype=np.float64)
    assert v.shape == (n,), s
    return v


def main():
    root = ET.parse(SRC).getroot()
    world = root.find("worldbody")
    names, parent, pos, mass, ipos, tensor, diag = [], [], [], [], [], [], []
    jname, jbody, jaxis, jrange = [], [], [], []
    gbody, gtype, gsize, gpos = [], [], [], []

    def geoms(elem, b):
        for g in elem.findall("geom"):
            t = g.get("type", "sphere")
            if t == "mesh":
                continue  # visual only (contype = conaffinity = 0)
            size = [float(s) for s in g.get("size").split()]
            gbody.append(b); gtype.append(GEOM_CODE[t]); gsize.append((size + [0.0, 0.0, 0.0])[:3]); gpos.append(vec(g.get("pos", "0 0 0"), 3))
            assert g.get("quat") is None

    def visit(elem, par):
        b = len(names)
        names.append(elem.get("name")); parent.append(par); pos.append(vec(elem.get("pos", "0 0 0"), 3))
        assert elem.get("quat") is None and elem.get("euler") is None
        ine = elem.find("inertial")
        mass.append(float(ine.get("mass"))); ipos.append(vec(ine.get("pos"), 3))
        R, d = quat_to_mat(vec(ine.get("quat", "1 0 0 0"), 4)), vec(ine.get("diaginertia"), 3)
        tensor.append(R @ np.diag(d) @ R.T); diag.append(d)
        for j in elem.findall("joint"):
            if j.get("type", "hinge") == "free":
                continue
            assert vec(j.get("pos", "0 0 0"), 3).tolist() == [0, 0, 0] and j.get("limited") == "true"
            jname.append(j.get("name")); jbody.append(b); jaxis.append(vec(j.get("axis"), 3)); jrange.append(vec(j.get("range"), 2))
        geoms(elem, b)
        for c in elem.findall("body"):
            visit(c, b)

    geoms(world, -1)
    for bd in world.findall("body"):
        visit(bd, -1)
    motors = root.find("actuator").findall("motor")
    assert [m.get("joint") for m in motors] == jname
    ctrl = np.array([vec(m.get("ctrlrange"), 2) for m in motors])
    opt = root.find("option")
    np.savez_compressed(
        "tests/golden/mjcf_model.npz",
        body_names=np.array(names), parent=np.array(parent, dtype=np.int32), body_pos=np.array(pos), mass=np.array(mass), inertial_pos=np.array(ipos),
        inertia_tensor=np.array(tensor), diaginertia=np.array(diag),
        joint_names=np.array(jname), joint_body=np.array(jbody, dtype=np.int32), joint_axis=np.array(jaxis), joint_range=np.array(jrange), ctrlrange=ctrl,
        geom_body=np.array(gbody, dtype=np.int32), geom_type=np.array(gtype, dtype=np.int32), geom_halfsize=np.array(gsize), geom_pos=np.array(gpos),
        timestep=np.array(float(opt.get("timestep")) if opt is not None and opt.get("timestep") else np.nan))
    print(len(names), "bodies", len(jname), "hinges", len(gbody), "collision geoms; total mass", sum(mass))


if __name__ == "__main__":
    main()
