"""ctypes front-end of oracle/dyn_ref.c -- TEST INFRASTRUCTURE (see the C file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NB, ND = 13, 12


def _structs(ct):
    """ctypes mirrors of ref_model_t / ref_phys_t / ref_terrain_t for one real type (c_double: libdynref.so, c_float: libdynref32.so)."""

    class RefModel(C.Structure):
        _fields_ = [
            ("nb", C.c_int32),
            ("parent", C.c_int32 * NB),
            ("axis", C.c_int32 * NB),
            ("pos", ct * 3 * NB),
            ("mass", ct * NB),
            ("com", ct * 3 * NB),
            ("inertia", ct * 6 * NB),
            ("q_lower", ct * ND),
            ("q_upper", ct * ND),
            ("qd_limit", ct * ND),
            ("foot_body", C.c_int32 * 2),
            ("foot_corner", ct * 3 * 4),
            ("n_sph", C.c_int32),
            ("sph_body", C.c_int32 * 16),
            ("sph_pos", ct * 3 * 16),
            ("sph_r", ct * 16),
            ("n_cap", C.c_int32),
            ("cap_body", C.c_int32 * 4),
            ("cap_a", ct * 3 * 4),
            ("cap_b", ct * 3 * 4),
            ("cap_r", ct * 4),
        ]

    class RefPhys(C.Structure):
        _fields_ = [
            ("dt", ct),
            ("g", ct * 3),
            ("contact_k", ct),
            ("contact_d", ct),
            ("contact_ramp", ct),
            ("friction_visc", ct),
            ("limit_k", ct),
            ("limit_d", ct),
            ("terrain_mu", ct),
            ("terrain_restitution", ct),
            ("clamp_qd", C.c_int32),
            ("pad", C.c_int32),
            ("body_gate_height", ct),
            ("self_k", ct),
            ("self_d", ct),
            ("self_mu", ct),
            ("self_visc", ct),
            ("self_collisions", C.c_int32),
            ("pad2", C.c_int32),
        ]

    class RefTerrain(C.Structure):
        _fields_ = [
            ("type", C.c_int32),
            ("rows", C.c_int32),
            ("cols", C.c_int32),
            ("border_px", C.c_int32),
            ("hscale", ct),
            ("vscale", ct),
            ("hf", C.c_void_p),
        ]

    return RefModel, RefPhys, RefTerrain


RefModel, RefPhys, RefTerrain = _structs(C.c_double)
_STRUCTS32 = _structs(C.c_float)


def build(force=False, f32=False):
    name = "libdynref32.so" if f32 else "libdynref.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "dyn_ref.c")
    if os.environ.get("BG_SANITIZE", "0") == "1":  # tests/test_sanitizers.py: the same source under the address / undefined-behaviour sanitizers
        import tempfile

        so = os.path.join(tempfile.mkdtemp(prefix="bg_san_"), name)
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-fopenmp", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
                              + (["-DREF_F32", "-fsingle-precision-constant"] if f32 else []) + ["-shared", "-o", so, src, "-lm"])
        return so
    if force or not os.path.isfile(so) or (os.path.isfile(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", name])
    return so


_libs = {}


def lib(f32=False):
    if f32 not in _libs:
        l = C.CDLL(build(f32=f32))
        ct = C.c_float if f32 else C.c_double
        l.ref_terrain_height.restype = ct
        l.ref_terrain_height.argtypes = [C.c_void_p, ct, ct]
        _libs[f32] = l
    return _libs[f32]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


# default physics constants of this build; product defaults must match
DEFAULT_PHYS = dict(
    dt=0.002, g=(0.0, 0.0, -9.81), contact_k=4.0e4, contact_d=600.0, contact_ramp=1.0e-3, friction_visc=1.0e4,
    limit_k=2000.0, limit_d=20.0, terrain_mu=1.0, terrain_restitution=0.0, clamp_qd=1, body_gate_height=0.45,
    # leg against leg (asset.self_collisions: 0 = on, envs/T1.yaml:69): explicit penalty between the shank / foot capsules of the two legs
    self_collisions=1, self_k=4.0e4, self_d=150.0, self_mu=1.0, self_visc=100.0,
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


def self_capsules(flat_model, foot_bodies):
    """Self-collision capsules [(body, a, b, radius)] in link coordinates, per leg the shank then the foot (URDF <collision>,
    resources/T1/T1_locomotion.xml:71,80,104,113).  Shank (the link two above the foot): the capsule inscribed in its z-axis cylinder, i.e. the
    segment between the centres of the two end spheres that also meet the terrain.  Foot box (length L >= width W >= height H, L along x): radius
    W / 2 (the box's half width, so feet side by side touch where the boxes do), segment of half length L / 2 - H about the box centre: the
    round ends start one sole thickness inside the box ends and overhang them by W / 2 - H on the centre line (2 cm for the T1)."""
    out = []
    for fb in foot_bodies:
        shank = int(flat_model.parent[int(flat_model.parent[fb])])
        cyl = [sh for sh in flat_model.shapes if int(sh["body"]) == shank and sh["type"] == "cylinder"]
        box = [sh for sh in flat_model.shapes if int(sh["body"]) == fb and sh["type"] == "box"]
        if len(cyl) != 1 or len(box) != 1:
            raise ValueError("self-collision capsules: expected one cylinder on each shank and one box on each foot")
        pos, (r, length) = [float(v) for v in cyl[0]["pos"]], [float(v) for v in cyl[0]["size"]]
        h = max(0.5 * length - r, 0.0)
        out.append((shank, (pos[0], pos[1], pos[2] - h), (pos[0], pos[1], pos[2] + h), r))
        pos, (lx, ly, lz) = [float(v) for v in box[0]["pos"]], [float(v) for v in box[0]["size"]]
        if not (lx >= ly >= lz):
            raise ValueError("self-collision capsules: the foot box must be longest along x and thinnest along z")
        h = max(0.5 * lx - lz, 0.0)
        out.append((fb, (pos[0] - h, pos[1], pos[2]), (pos[0] + h, pos[1], pos[2]), 0.5 * ly))
    return out


class DynRef:
    """Holds model/physics/terrain structs and exposes the oracle entry points on numpy arrays."""

    def __init__(self, flat_model, foot_names=("left_foot_link", "right_foot_link"), feet_edge_pos=None, phys=None, terrain=None, body_contacts=True,
                 real="f64"):
        """real: "f64" = the oracle proper (libdynref.so); "f32" = the single-precision twin of the same source (libdynref32.so), used as the
        per-state measure of fp32 rounding sensitivity and as the CPU baseline.  Inputs / outputs are float64 numpy arrays either way."""
        self.f32 = real == "f32"
        self.rt = np.float32 if self.f32 else np.float64
        self._S = _STRUCTS32 if self.f32 else (RefModel, RefPhys, RefTerrain)
        m = self._S[0]()
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
        caps = self_capsules(flat_model, [m.foot_body[0], m.foot_body[1]])
        m.n_cap = len(caps)
        for k, (b, a, bb, r) in enumerate(caps):
            m.cap_body[k] = b
            m.cap_r[k] = r
            for ax in range(3):
                m.cap_a[k][ax] = a[ax]
                m.cap_b[k][ax] = bb[ax]
        self.model = m
        ph = dict(DEFAULT_PHYS)
        ph.update(phys or {})
        p = self._S[1]()
        p.dt = ph["dt"]
        for a in range(3):
            p.g[a] = ph["g"][a]
        for k in ("contact_k", "contact_d", "contact_ramp", "friction_visc", "limit_k", "limit_d", "terrain_mu", "terrain_restitution"):
            setattr(p, k, float(ph[k]))
        p.clamp_qd = int(ph["clamp_qd"])
        p.body_gate_height = float(ph["body_gate_height"])
        for k in ("self_k", "self_d", "self_mu", "self_visc"):
            setattr(p, k, float(ph[k]))
        p.self_collisions = int(ph["self_collisions"])
        self.phys = p
        self.set_terrain(terrain)

    def set_terrain(self, terrain):
        """terrain: None/plane, or dict(height_field_raw int16[rows,cols], hscale, vscale, border_px)."""
        t = self._S[2]()
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

    # ---- array plumbing: callers hand float64 numpy arrays; the f32 twin gets float32 copies and results are widened back
    def _in(self, a):
        return None if a is None else np.ascontiguousarray(a, dtype=self.rt)

    def _out(self, *shape):
        return np.zeros(shape, dtype=self.rt)

    def _wide(self, a):
        return None if a is None else (a.astype(np.float64) if self.f32 else a)

    def _inplace(self, arrays):
        """Working copies of in-place float64 arrays in the library's real type (the arrays themselves for the double build)."""
        for a in arrays:
            assert a.dtype == np.float64 and a.flags.c_contiguous
        return [a.astype(np.float32) for a in arrays] if self.f32 else list(arrays)

    def _writeback(self, arrays, work):
        if self.f32:
            for a, w in zip(arrays, work):
                a[...] = w

    @property
    def _l(self):
        return lib(self.f32)

    def _refs(self):
        return C.byref(self.model), C.byref(self.phys), C.byref(self.terrain)

    def terrain_height(self, x, y):
        return float(self._l.ref_terrain_height(C.byref(self.terrain), float(x), float(y)))

    def segment_closest(self, a1, b1, a2, b2):
        """(s, t) of the regularised closest points of two segments (the self-collision narrow phase)."""
        seg = self._in(np.concatenate([a1, b1, a2, b2]))
        st = self._out(2)
        self._l.ref_segment_closest(_p(seg), _p(st))
        return float(st[0]), float(st[1])

    def self_contact_forces(self, root, q, qd):
        """World-frame forces [13,3] of the leg-against-leg contacts alone in the given state."""
        out = self._out(NB, 3)
        self._l.ref_self_contact_forces(C.byref(self.model), C.byref(self.phys), _p(self._in(root)), _p(self._in(q)), _p(self._in(qd)), _p(out))
        return self._wide(out)

    def forward(self, root, q, qd, tau, base_wrench=None, mass_scale=None, com_off=None, foot_mat=None, want_body_acc=False):
        i = self._in
        qacc, cf = self._out(18), self._out(NB, 3)
        ab = self._out(NB, 6) if want_body_acc else None
        r = self._l.ref_forward(*self._refs(), _p(i(mass_scale)), _p(i(com_off)), _p(i(foot_mat)), _p(i(root)), _p(i(q)), _p(i(qd)), _p(i(tau)),
                                _p(i(base_wrench)), _p(qacc), _p(cf), _p(ab))
        if r:
            raise RuntimeError("ref_forward failed")
        return (self._wide(qacc), self._wide(cf), self._wide(ab)) if want_body_acc else (self._wide(qacc), self._wide(cf))

    def inverse(self, root, q, qd, qacc, mass_scale=None, com_off=None):
        i = self._in
        res = self._out(18)
        self._l.ref_inverse(C.byref(self.model), C.byref(self.phys), _p(i(mass_scale)), _p(i(com_off)), _p(i(root)), _p(i(q)), _p(i(qd)), _p(i(qacc)), _p(res))
        return self._wide(res)

    def body_poses(self, root, q):
        pos, rot = self._out(NB, 3), self._out(NB, 3, 3)
        self._l.ref_body_poses(C.byref(self.model), _p(self._in(root)), _p(self._in(q)), _p(pos), _p(rot))
        return self._wide(pos), self._wide(rot)

    def step(self, root, q, qd, tau, base_wrench=None, mass_scale=None, com_off=None, foot_mat=None):
        """One substep, in place on float64 arrays root[13], q[12], qd[12]. Returns contact forces [13,3]."""
        i = self._in
        st = (root, q, qd)
        w = self._inplace(st)
        cf = self._out(NB, 3)
        r = self._l.ref_step(*self._refs(), _p(i(mass_scale)), _p(i(com_off)), _p(i(foot_mat)), _p(w[0]), _p(w[1]), _p(w[2]), _p(i(tau)),
                             _p(i(base_wrench)), _p(cf))
        if r:
            raise RuntimeError("ref_step failed")
        self._writeback(st, w)
        return self._wide(cf)

    def forward_bw(self, root, q, qd, tau, body_force=None, body_torque=None, mass_scale=None, com_off=None, foot_mat=None):
        """Forward dynamics with a force / torque on every body (local frame, force at the centre of mass): t1.py:522-527."""
        i = self._in
        qacc, cf = self._out(18), self._out(NB, 3)
        r = self._l.ref_forward_bw(*self._refs(), _p(i(mass_scale)), _p(i(com_off)), _p(i(foot_mat)), _p(i(root)), _p(i(q)), _p(i(qd)), _p(i(tau)),
                                   _p(i(body_force)), _p(i(body_torque)), _p(qacc), _p(cf))
        if r:
            raise RuntimeError("ref_forward_bw failed")
        return self._wide(qacc), self._wide(cf)

    def step_bw(self, root, q, qd, tau, body_force=None, body_torque=None, mass_scale=None, com_off=None, foot_mat=None):
        """One gym.simulate (t1.py:451) in place on float64 root[13], q[12], qd[12] with per-body applied forces. Returns contact forces."""
        i = self._in
        st = (root, q, qd)
        w = self._inplace(st)
        cf = self._out(NB, 3)
        r = self._l.ref_step_bw(*self._refs(), _p(i(mass_scale)), _p(i(com_off)), _p(i(foot_mat)), _p(w[0]), _p(w[1]), _p(w[2]), _p(i(tau)),
                                _p(i(body_force)), _p(i(body_torque)), _p(cf))
        if r:
            raise RuntimeError("ref_step_bw failed")
        self._writeback(st, w)
        return self._wide(cf)

    def body_states(self, root, q, qd):
        """Rigid-body state rows [13,13] (t1.py:220): pos, quat xyzw (w >= 0), lin vel of the origin, ang vel, world frame."""
        out = self._out(NB, 13)
        self._l.ref_body_states(C.byref(self.model), _p(self._in(root)), _p(self._in(q)), _p(self._in(qd)), _p(out))
        return self._wide(out)

    def body_states_batch(self, root, q, qd):
        """body_states for n envs at once: [n,13,13]."""
        n = root.shape[0]
        out = self._out(n, NB, 13)
        self._l.ref_body_states_batch(C.byref(self.model), int(n), _p(self._in(root)), _p(self._in(q)), _p(self._in(qd)), _p(out))
        return self._wide(out)

    def substeps_batch(self, decimation, mass_scale, com_off, foot_mat, kp, kd, fric, tau_limit, root, q, qd, targets, last_targets,
                       delay, base_wrench):
        """Decimation loop for n envs, in place on root[n,13], q[n,12], qd[n,12], last_targets[n,12] (float64)."""
        n = root.shape[0]
        i = self._in
        st = (root, q, qd, last_targets)
        w = self._inplace(st)
        delay = np.ascontiguousarray(delay, dtype=np.int32)
        tm, cf = self._out(n, ND), self._out(n, NB, 3)
        r = self._l.ref_substeps_batch(*self._refs(), int(decimation), int(n), _p(i(mass_scale)), _p(i(com_off)), _p(i(foot_mat)), _p(i(kp)), _p(i(kd)),
                                       _p(i(fric)), _p(i(tau_limit)), _p(w[0]), _p(w[1]), _p(w[2]), _p(i(targets)), _p(w[3]), _p(delay), _p(i(base_wrench)),
                                       _p(tm), _p(cf))
        if r:
            raise RuntimeError("ref_substeps_batch failed")
        self._writeback(st, w)
        return self._wide(tm), self._wide(cf)
