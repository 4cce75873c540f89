"""ctypes front-end of oracle/dyn_ref.c -- TEST INFRASTRUCTURE (see the C file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NB, ND = 13, 12


class RefModel(C.Structure):
    _fields_ = [
        ("nb", C.c_int32),
        ("parent", C.c_int32 * NB),
        ("axis", C.c_int32 * NB),
        ("pos", C.c_double * 3 * NB),
        ("mass", C.c_double * NB),
        ("com", C.c_double * 3 * NB),
        ("inertia", C.c_double * 6 * NB),
        ("q_lower", C.c_double * ND),
        ("q_upper", C.c_double * ND),
        ("qd_limit", C.c_double * ND),
        ("foot_body", C.c_int32 * 2),
        ("foot_corner", C.c_double * 3 * 4),
        ("n_sph", C.c_int32),
        ("sph_body", C.c_int32 * 16),
        ("sph_pos", C.c_double * 3 * 16),
        ("sph_r", C.c_double * 16),
    ]


class RefPhys(C.Structure):
    _fields_ = [
        ("dt", C.c_double),
        ("g", C.c_double * 3),
        ("contact_k", C.c_double),
        ("contact_d", C.c_double),
        ("contact_ramp", C.c_double),
        ("friction_visc", C.c_double),
        ("limit_k", C.c_double),
        ("limit_d", C.c_double),
        ("terrain_mu", C.c_double),
        ("terrain_restitution", C.c_double),
        ("clamp_qd", C.c_int32),
        ("pad", C.c_int32),
        ("body_gate_height", C.c_double),
    ]


class RefTerrain(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("border_px", C.c_int32),
        ("hscale", C.c_double),
        ("vscale", C.c_double),
        ("hf", C.c_void_p),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libdynref.so")
    src = os.path.join(_HERE, "dyn_ref.c")
    if force or not os.path.isfile(so) or (os.path.isfile(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libdynref.so"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ref_terrain_height.restype = C.c_double
        _lib.ref_terrain_height.argtypes = [C.c_void_p, C.c_double, C.c_double]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


# default physics constants of this build; product defaults must match
DEFAULT_PHYS = dict(
    dt=0.002, g=(0.0, 0.0, -9.81), contact_k=4.0e4, contact_d=600.0, contact_ramp=1.0e-3, friction_visc=1.0e4,
    limit_k=2000.0, limit_d=20.0, terrain_mu=1.0, terrain_restitution=0.0, clamp_qd=1, body_gate_height=0.45,
)


def contact_spheres(flat_model, foot_bodies):
    """Contact spheres of the non-foot collision shapes (URDF <collision>, resources/T1/T1_locomotion.xml:42,66,71,99,104): the 8 corners of a
    box with radius 0, two spheres inscribed in the ends of a cylinder (axis z).  Returns [(body, (x, y, z), radius)]."""
    out = []
    for sh in flat_model.shapes:
        b, pos, size = int(sh["body"]), [float(v) for v in sh["pos"]], [float(v) for v in sh["size"]]
        if b in foot_bodies:
            continue
        if sh["type"] == "box":
            for sx in (-0.5, 0.5):
                for sy in (-0.5, 0.5):
                    for sz in (-0.5, 0.5):
                        out.append((b, (pos[0] + sx * size[0], pos[1] + sy * size[1], pos[2] + sz * size[2]), 0.0))
        elif sh["type"] == "cylinder":
            r, length = size[0], size[1]
            for sz in (-1.0, 1.0):
                out.append((b, (pos[0], pos[1], pos[2] + sz * max(0.5 * length - r, 0.0)), r))
    return out


class DynRef:
    """Holds model/physics/terrain structs and exposes the oracle entry points on numpy arrays."""

    def __init__(self, flat_model, foot_names=("left_foot_link", "right_foot_link"), feet_edge_pos=None, phys=None, terrain=None, body_contacts=True):
        m = RefModel()
        m.nb = flat_model.num_bodies
        assert m.nb == NB and flat_model.num_dofs == ND
        for i in range(NB):
            m.parent[i] = int(flat_model.parent[i])
            m.axis[i] = int(flat_model.joint_axis[i])
            m.mass[i] = float(flat_model.mass[i])
            for a in range(3):
                m.pos[i][a] = float(flat_model.body_pos[i, a])
                m.com[i][a] = float(flat_model.com[i, a])
            for a in range(6):
                m.inertia[i][a] = float(flat_model.inertia[i, a])
        for j in range(ND):
            m.q_lower[j] = float(flat_model.dof_lower[j])
            m.q_upper[j] = float(flat_model.dof_upper[j])
            m.qd_limit[j] = float(flat_model.dof_velocity[j])
        for f in range(2):
            m.foot_body[f] = flat_model.find_body(foot_names[f])
        if feet_edge_pos is None:
            feet_edge_pos = [[0.1215, 0.05, -0.03], [0.1215, -0.05, -0.03], [-0.1015, 0.05, -0.03], [-0.1015, -0.05, -0.03]]
        for c in range(4):
            for a in range(3):
                m.foot_corner[c][a] = float(feet_edge_pos[c][a])
        sph = contact_spheres(flat_model, [m.foot_body[0], m.foot_body[1]]) if body_contacts else []
        assert len(sph) <= 16
        m.n_sph = len(sph)
        for k, (b, c, r) in enumerate(sph):
            m.sph_body[k] = b
            m.sph_r[k] = r
            for a in range(3):
                m.sph_pos[k][a] = c[a]
        self.model = m
        ph = dict(DEFAULT_PHYS)
        ph.update(phys or {})
        p = RefPhys()
        p.dt = ph["dt"]
        for a in range(3):
            p.g[a] = ph["g"][a]
        for k in ("contact_k", "contact_d", "contact_ramp", "friction_visc", "limit_k", "limit_d", "terrain_mu", "terrain_restitution"):
            setattr(p, k, float(ph[k]))
        p.clamp_qd = int(ph["clamp_qd"])
        p.body_gate_height = float(ph["body_gate_height"])
        self.phys = p
        self.set_terrain(terrain)

    def set_terrain(self, terrain):
        """terrain: None/plane, or dict(height_field_raw int16[rows,cols], hscale, vscale, border_px)."""
        t = RefTerrain()
        if terrain is None:
            t.type = 0
            self._hf = None
        else:
            self._hf = np.ascontiguousarray(terrain["height_field_raw"], dtype=np.int16)
            t.type = 1
            t.rows, t.cols = self._hf.shape
            t.border_px = int(terrain["border_px"])
            t.hscale = float(terrain["hscale"])
            t.vscale = float(terrain["vscale"])
            t.hf = self._hf.ctypes.data
        self.terrain = t

    def terrain_height(self, x, y):
        return lib().ref_terrain_height(C.byref(self.terrain), float(x), float(y))

    def forward(self, root, q, qd, tau, base_wrench=None, mass_scale=None, com_off=None, foot_mat=None, want_body_acc=False):
        root, q, qd, tau = _f64(root), _f64(q), _f64(qd), _f64(tau)
        base_wrench, mass_scale, com_off, foot_mat = _f64(base_wrench), _f64(mass_scale), _f64(com_off), _f64(foot_mat)
        qacc = np.zeros(18)
        cf = np.zeros((NB, 3))
        ab = np.zeros((NB, 6)) if want_body_acc else None
        r = lib().ref_forward(C.byref(self.model), C.byref(self.phys), C.byref(self.terrain), _p(mass_scale), _p(com_off), _p(foot_mat),
                              _p(root), _p(q), _p(qd), _p(tau), _p(base_wrench), _p(qacc), _p(cf), _p(ab))
        if r:
            raise RuntimeError("ref_forward failed")
        return (qacc, cf, ab) if want_body_acc else (qacc, cf)

    def inverse(self, root, q, qd, qacc, mass_scale=None, com_off=None):
        root, q, qd, qacc = _f64(root), _f64(q), _f64(qd), _f64(qacc)
        mass_scale, com_off = _f64(mass_scale), _f64(com_off)
        res = np.zeros(18)
        lib().ref_inverse(C.byref(self.model), C.byref(self.phys), _p(mass_scale), _p(com_off), _p(root), _p(q), _p(qd), _p(qacc), _p(res))
        return res

    def body_poses(self, root, q):
        root, q = _f64(root), _f64(q)
        pos, rot = np.zeros((NB, 3)), np.zeros((NB, 3, 3))
        lib().ref_body_poses(C.byref(self.model), _p(root), _p(q), _p(pos), _p(rot))
        return pos, rot

    def step(self, root, q, qd, tau, base_wrench=None, mass_scale=None, com_off=None, foot_mat=None):
        """One substep, in place on float64 arrays root[13], q[12], qd[12]. Returns contact forces [13,3]."""
        assert root.dtype == np.float64 and q.dtype == np.float64 and qd.dtype == np.float64
        tau, base_wrench, mass_scale, com_off, foot_mat = _f64(tau), _f64(base_wrench), _f64(mass_scale), _f64(com_off), _f64(foot_mat)
        cf = np.zeros((NB, 3))
        r = lib().ref_step(C.byref(self.model), C.byref(self.phys), C.byref(self.terrain), _p(mass_scale), _p(com_off), _p(foot_mat),
                           _p(root), _p(q), _p(qd), _p(tau), _p(base_wrench), _p(cf))
        if r:
            raise RuntimeError("ref_step failed")
        return cf

    def forward_bw(self, root, q, qd, tau, body_force=None, body_torque=None, mass_scale=None, com_off=None, foot_mat=None):
        """Forward dynamics with a force / torque on every body (local frame, force at the centre of mass): t1.py:522-527."""
        root, q, qd, tau = _f64(root), _f64(q), _f64(qd), _f64(tau)
        body_force, body_torque, mass_scale, com_off, foot_mat = map(_f64, (body_force, body_torque, mass_scale, com_off, foot_mat))
        qacc = np.zeros(18)
        cf = np.zeros((NB, 3))
        r = lib().ref_forward_bw(C.byref(self.model), C.byref(self.phys), C.byref(self.terrain), _p(mass_scale), _p(com_off), _p(foot_mat),
                                 _p(root), _p(q), _p(qd), _p(tau), _p(body_force), _p(body_torque), _p(qacc), _p(cf))
        if r:
            raise RuntimeError("ref_forward_bw failed")
        return qacc, cf

    def step_bw(self, root, q, qd, tau, body_force=None, body_torque=None, mass_scale=None, com_off=None, foot_mat=None):
        """One gym.simulate (t1.py:451) in place on float64 root[13], q[12], qd[12] with per-body applied forces. Returns contact forces."""
        assert root.dtype == np.float64 and q.dtype == np.float64 and qd.dtype == np.float64
        tau, body_force, body_torque, mass_scale, com_off, foot_mat = map(_f64, (tau, body_force, body_torque, mass_scale, com_off, foot_mat))
        cf = np.zeros((NB, 3))
        r = lib().ref_step_bw(C.byref(self.model), C.byref(self.phys), C.byref(self.terrain), _p(mass_scale), _p(com_off), _p(foot_mat),
                              _p(root), _p(q), _p(qd), _p(tau), _p(body_force), _p(body_torque), _p(cf))
        if r:
            raise RuntimeError("ref_step_bw failed")
        return cf

    def body_states(self, root, q, qd):
        """Rigid-body state rows [13,13] (t1.py:220): pos, quat xyzw (w >= 0), lin vel of the origin, ang vel, world frame."""
        root, q, qd = _f64(root), _f64(q), _f64(qd)
        out = np.zeros((NB, 13))
        lib().ref_body_states(C.byref(self.model), _p(root), _p(q), _p(qd), _p(out))
        return out

    def substeps_batch(self, decimation, mass_scale, com_off, foot_mat, kp, kd, fric, tau_limit, root, q, qd, targets, last_targets,
                       delay, base_wrench):
        """Decimation loop for n envs, in place on root[n,13], q[n,12], qd[n,12], last_targets[n,12] (float64)."""
        n = root.shape[0]
        for a in (root, q, qd, last_targets):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        mass_scale, com_off, foot_mat, kp, kd, fric = map(_f64, (mass_scale, com_off, foot_mat, kp, kd, fric))
        tau_limit, targets, base_wrench = _f64(tau_limit), _f64(targets), _f64(base_wrench)
        delay = np.ascontiguousarray(delay, dtype=np.int32)
        tm = np.zeros((n, ND))
        cf = np.zeros((n, NB, 3))
        r = lib().ref_substeps_batch(C.byref(self.model), C.byref(self.phys), C.byref(self.terrain), int(decimation), int(n),
                                     _p(mass_scale), _p(com_off), _p(foot_mat), _p(kp), _p(kd), _p(fric), _p(tau_limit), _p(root), _p(q),
                                     _p(qd), _p(targets), _p(last_targets), _p(delay), _p(base_wrench), _p(tm), _p(cf))
        if r:
            raise RuntimeError("ref_substeps_batch failed")
        return tm, cf
