"""URDF -> flat articulated-body model (host side, start-up only).

Replaces what the reference obtains from Isaac Gym's asset loader
(`envs/t1.py:39-59` load_asset with `collapse_fixed_joints`, `:55-67`
dof/body queries, `:85-108` body-index lookups).  The result is a plain
numeric description of the 13-body / 12-DoF tree that is handed to the
native library through `bg_model_create` (include/booster_gym_amd.h).

Conventions of the flat model
  * bodies in depth-first URDF order (Trunk, left leg 1..6, right leg 7..12)
  * `body_pos[i]`   origin of body i in its parent frame (joint origin xyz)
  * `joint_axis[i]` 0 = floating base, 1/2/3 = revolute about +x/+y/+z
  * `mass, com, inertia` of each body in its own frame; `inertia` is about the
    centre of mass, order (xx, yy, zz, xy, xz, yz)
"""

from __future__ import annotations

import json
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    if len(v) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return np.array(v, dtype=np.float64)


def _rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _origin(elem):
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.zeros(3), np.eye(3)
    xyz = _vec(o.get("xyz", "0 0 0"))
    rpy = _vec(o.get("rpy", "0 0 0"))
    return xyz, _rpy_to_mat(rpy)


@dataclass
class _Link:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about com, link axes
    shapes: list = field(default_factory=list)


def _merge(parent: _Link, child: _Link, xyz, rot):
    """Rigidly attach `child` (pose xyz/rot in parent frame) to `parent`."""
    c_com = xyz + rot @ child.com
    c_in = rot @ child.inertia @ rot.T
    m = parent.mass + child.mass
    if m <= 0.0:
        return
    com = (parent.mass * parent.com + child.mass * c_com) / m

    def shift(inertia, mass, d):
        return inertia + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))

    parent.inertia = shift(parent.inertia, parent.mass, parent.com - com) + shift(c_in, child.mass, c_com - com)
    parent.mass, parent.com = m, com
    for s in child.shapes:
        s2 = dict(s)
        s2["pos"] = xyz + rot @ s["pos"]
        s2["rot"] = rot @ s["rot"]
        parent.shapes.append(s2)


@dataclass
class FlatModel:
    name: str
    body_names: list
    parent: np.ndarray  # int32 [nb]
    joint_axis: np.ndarray  # int32 [nb]
    body_pos: np.ndarray  # f64 [nb,3]
    mass: np.ndarray  # f64 [nb]
    com: np.ndarray  # f64 [nb,3]
    inertia: np.ndarray  # f64 [nb,6]
    dof_names: list
    dof_lower: np.ndarray
    dof_upper: np.ndarray
    dof_velocity: np.ndarray
    dof_effort: np.ndarray
    shapes: list  # dicts: body, type, size, pos

    @property
    def num_bodies(self):
        return len(self.body_names)

    @property
    def num_dofs(self):
        return len(self.dof_names)

    def find_body(self, name):
        return self.body_names.index(name)

    def contact_spheres(self, exclude_bodies=()):
        """Contact spheres (body, centre, radius) standing in for the collision primitives of the bodies NOT in `exclude_bodies` (the feet have
        their own sole-corner contacts): a box gives its 8 corners with radius 0, a z-axis cylinder two spheres of its radius inscribed in
        its ends.  Sorted by body index."""
        out = []
        for sh in self.shapes:
            b = int(sh["body"])
            if b in exclude_bodies:
                continue
            px, py, pz = (float(v) for v in sh["pos"])
            if sh["type"] == "box":
                hx, hy, hz = (0.5 * float(v) for v in sh["size"])
                out += [(b, (px + i * hx, py + j * hy, pz + k * hz), 0.0) for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)]
            elif sh["type"] == "cylinder":
                r, half = float(sh["size"][0]), 0.5 * float(sh["size"][1])
                out += [(b, (px, py, pz + sgn * max(half - r, 0.0)), r) for sgn in (-1.0, 1.0)]
        return sorted(out, key=lambda t: t[0])

    def self_collision_capsules(self, foot_bodies):
        """Self-collision capsules of the legs (reference: `self_collisions` passed to create_actor, envs/t1.py:128, envs/T1.yaml:69), per foot body
        `[shank, foot]`, each `(body, a, b, radius)` with the segment end points in link coordinates.  Shank = the link two above the foot: the
        capsule inscribed in its z-axis cylinder (the segment between the centres of the end spheres that `contact_spheres` returns).  Foot box
        (L >= W >= H along x, y, z): radius W / 2, the box's half width, and half length L / 2 - H about the box centre, so that feet side by side
        touch where the boxes do and the round ends start one sole thickness inside the box ends."""
        out = []
        for fb in foot_bodies:
            shank = int(self.parent[int(self.parent[fb])])
            cyl = [sh for sh in self.shapes if int(sh["body"]) == shank and sh["type"] == "cylinder"]
            box = [sh for sh in self.shapes if int(sh["body"]) == int(fb) and sh["type"] == "box"]
            if len(cyl) != 1 or len(box) != 1:
                # an asset without that geometry: no self-collision (radius 0 = "no capsule" in bg_model_desc; bg_env_create then leaves the
                # leg-against-leg contacts off), as the C ABI documents -- not an error
                import warnings

                warnings.warn("self-collision capsules need one cylinder on each shank and one box on each foot (URDF <collision>): none derived, "
                              "leg-against-leg contacts are off for this asset")
                return []
            (px, py, pz), (r, length) = (float(v) for v in cyl[0]["pos"]), (float(v) for v in cyl[0]["size"])
            h = max(0.5 * length - r, 0.0)
            caps = [(shank, (px, py, pz - h), (px, py, pz + h), r)]
            (px, py, pz), (lx, ly, lz) = (float(v) for v in box[0]["pos"]), (float(v) for v in box[0]["size"])
            if not lx >= ly >= lz:
                raise ValueError("self-collision capsules: the foot box must be longest along x and thinnest along z")
            h = max(0.5 * lx - lz, 0.0)
            caps.append((int(fb), (px - h, py, pz), (px + h, py, pz), 0.5 * ly))
            out.append(caps)
        return out

    def to_json(self):
        return {
            "format": "booster_gym_amd.flat_model.v1",
            "name": self.name,
            "body_names": self.body_names,
            "parent": self.parent.tolist(),
            "joint_axis": self.joint_axis.tolist(),
            "body_pos": self.body_pos.tolist(),
            "mass": self.mass.tolist(),
            "com": self.com.tolist(),
            "inertia": self.inertia.tolist(),
            "dof_names": self.dof_names,
            "dof_lower": self.dof_lower.tolist(),
            "dof_upper": self.dof_upper.tolist(),
            "dof_velocity": self.dof_velocity.tolist(),
            "dof_effort": self.dof_effort.tolist(),
            "shapes": self.shapes,
        }

    def save(self, path):
        with open(path, "w") as f:
            json.dump(self.to_json(), f, indent=1)

    @staticmethod
    def load(path):
        with open(path) as f:
            d = json.load(f)
        if d.get("format") != "booster_gym_amd.flat_model.v1":
            raise ValueError(f"{path}: not a flat model file")
        return FlatModel(
            name=d["name"],
            body_names=list(d["body_names"]),
            parent=np.array(d["parent"], dtype=np.int32),
            joint_axis=np.array(d["joint_axis"], dtype=np.int32),
            body_pos=np.array(d["body_pos"], dtype=np.float64),
            mass=np.array(d["mass"], dtype=np.float64),
            com=np.array(d["com"], dtype=np.float64),
            inertia=np.array(d["inertia"], dtype=np.float64),
            dof_names=list(d["dof_names"]),
            dof_lower=np.array(d["dof_lower"], dtype=np.float64),
            dof_upper=np.array(d["dof_upper"], dtype=np.float64),
            dof_velocity=np.array(d["dof_velocity"], dtype=np.float64),
            dof_effort=np.array(d["dof_effort"], dtype=np.float64),
            shapes=list(d["shapes"]),
        )


def load_urdf(path, collapse_fixed_joints=True) -> FlatModel:
    """Parse a URDF and (optionally) fold links behind fixed joints into their parents."""
    root = ET.parse(path).getroot()
    links = {}
    order = []
    for le in root.findall("link"):
        lk = _Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            xyz, rot = _origin(ine)
            lk.mass = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            i = {k: float(it.get(k, "0")) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")}
            tensor = np.array([[i["ixx"], i["ixy"], i["ixz"]], [i["ixy"], i["iyy"], i["iyz"]], [i["ixz"], i["iyz"], i["izz"]]])
            lk.com = xyz
            lk.inertia = rot @ tensor @ rot.T
        for ce in le.findall("collision"):
            xyz, rot = _origin(ce)
            g = ce.find("geometry")
            if g.find("box") is not None:
                lk.shapes.append({"type": "box", "size": _vec(g.find("box").get("size")).tolist(), "pos": xyz, "rot": rot})
            elif g.find("cylinder") is not None:
                c = g.find("cylinder")
                lk.shapes.append({"type": "cylinder", "size": [float(c.get("radius")), float(c.get("length"))], "pos": xyz, "rot": rot})
            elif g.find("sphere") is not None:
                lk.shapes.append({"type": "sphere", "size": [float(g.find("sphere").get("radius"))], "pos": xyz, "rot": rot})
            # mesh collisions carry no primitive data: ignored
        links[lk.name] = lk
        order.append(lk.name)

    joints = []
    children = set()
    for je in root.findall("joint"):
        xyz, rot = _origin(je)
        j = {
            "name": je.get("name"),
            "type": je.get("type"),
            "parent": je.find("parent").get("link"),
            "child": je.find("child").get("link"),
            "xyz": xyz,
            "rot": rot,
            "axis": _vec(je.find("axis").get("xyz")) if je.find("axis") is not None else np.array([1.0, 0, 0]),
        }
        lim = je.find("limit")
        j["limit"] = {k: float(lim.get(k, "0")) for k in ("lower", "upper", "effort", "velocity")} if lim is not None else None
        joints.append(j)
        children.add(j["child"])
    roots = [n for n in order if n not in children]
    if len(roots) != 1:
        raise ValueError(f"URDF must have exactly one root link, found {roots}")

    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)

    if collapse_fixed_joints:
        # fold leaves first so chains of fixed joints accumulate correctly
        def fold(name):
            for j in list(by_parent.get(name, [])):
                fold(j["child"])
                if j["type"] == "fixed":
                    _merge(links[name], links[j["child"]], j["xyz"], j["rot"])
                    by_parent[name].remove(j)
                    # re-parent the grandchildren through the fixed transform
                    for gj in by_parent.pop(j["child"], []):
                        gj = dict(gj)
                        gj["xyz"] = j["xyz"] + j["rot"] @ gj["xyz"]
                        gj["rot"] = j["rot"] @ gj["rot"]
                        gj["axis"] = gj["axis"]
                        gj["parent"] = name
                        by_parent.setdefault(name, []).append(gj)
        fold(roots[0])

    body_names, parent, axis, pos, dof = [], [], [], [], []

    def visit(name, par, j):
        idx = len(body_names)
        body_names.append(name)
        parent.append(par)
        if j is None:
            axis.append(0)
            pos.append(np.zeros(3))
        else:
            if j["type"] not in ("revolute", "continuous"):
                raise ValueError(f"joint {j['name']}: type {j['type']} unsupported (revolute / fixed only)")
            if not np.allclose(j["rot"], np.eye(3), atol=1e-9):
                raise ValueError(f"joint {j['name']}: rotated joint frames are unsupported")
            a = j["axis"]
            k = int(np.argmax(np.abs(a)))
            e = np.zeros(3)
            e[k] = 1.0
            if not np.allclose(a, e, atol=1e-9):
                raise ValueError(f"joint {j['name']}: axis {a} must be +x, +y or +z")
            axis.append(k + 1)
            pos.append(j["xyz"])
            dof.append(j)
        # keep URDF order of child joints (Isaac Gym orders DoFs depth-first: t1.py:57)
        for cj in sorted(by_parent.get(name, []), key=lambda q: order.index(q["child"])):
            visit(cj["child"], idx, cj)

    visit(roots[0], -1, None)

    nb = len(body_names)
    inertia6 = np.zeros((nb, 6))
    shapes = []
    for b, n in enumerate(body_names):
        t = links[n].inertia
        inertia6[b] = [t[0, 0], t[1, 1], t[2, 2], t[0, 1], t[0, 2], t[1, 2]]
        for s in links[n].shapes:
            if not np.allclose(s["rot"], np.eye(3), atol=1e-9):
                raise ValueError(f"link {n}: rotated collision primitives are unsupported")
            shapes.append({"body": b, "type": s["type"], "size": list(s["size"]), "pos": np.asarray(s["pos"]).tolist()})
    return FlatModel(
        name=root.get("name", "robot"),
        body_names=body_names,
        parent=np.array(parent, dtype=np.int32),
        joint_axis=np.array(axis, dtype=np.int32),
        body_pos=np.array(pos),
        mass=np.array([links[n].mass for n in body_names]),
        com=np.array([links[n].com for n in body_names]),
        inertia=inertia6,
        dof_names=[j["name"] for j in dof],
        dof_lower=np.array([j["limit"]["lower"] for j in dof]),
        dof_upper=np.array([j["limit"]["upper"] for j in dof]),
        dof_velocity=np.array([j["limit"]["velocity"] for j in dof]),
        dof_effort=np.array([j["limit"]["effort"] for j in dof]),
        shapes=shapes,
    )


def load_model(asset_file, collapse_fixed_joints=True) -> FlatModel:
    """Resolve `cfg["asset"]["file"]` (reference key, `envs/T1.yaml:61`).

    Search order: the path as given (cwd-relative, like the reference), the
    same path inside this package, then the packaged flat model next to it
    (`<stem>.flat.json`, numbers only) so that a box without the URDF still runs.
    """
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cands = [asset_file, os.path.join(here, asset_file)]
    for c in cands:
        if os.path.isfile(c):
            if c.endswith(".json"):
                return FlatModel.load(c)
            return load_urdf(c, collapse_fixed_joints)
    stem = os.path.splitext(asset_file)[0] + ".flat.json"
    for c in (stem, os.path.join(here, stem)):
        if os.path.isfile(c):
            return FlatModel.load(c)
    raise FileNotFoundError(f"robot asset not found: {asset_file} (also tried {stem})")
