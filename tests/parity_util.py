"""Step-parity bookkeeping shared by the GPU env tests (test infrastructure).

One env step of the fused HIP kernel is compared with the float64 oracle from IDENTICAL inputs.  Two requirements:
  bulk   over the whole test at least 1 - `max_explained_frac` (99 %) of the env-steps are within the stated tolerance on EVERY compared
         field at once (measured: 100 % on flat ground, 99.9 % on the rough terrain, the same fractions as the oracle's own fp32 build
         against its float64 build: profiles/r02_parity_diag_*.json, tools/parity_diag.py);
  hard   every env outside a tolerance must be EXPLAINED, otherwise the test fails.  Accepted explanations, each verified per env:
           contact_flag    a discrete flag of the final state differs (foot-contact flag; fields derived from it: feet_slip, feet_swing)
           limit_crossing  a joint sits within the position tolerance of its limit (dof_pos_limits counts it on one side only)
           fp32_backward   backward-error criterion: the float64 oracle, run from this env's inputs perturbed at fp32-rounding size
                           (relative 2e-6, then 2e-5), produces outcomes whose envelope contains the GPU's post-physics state.  A stiff
                           penalty contact near a switching surface (corner touching down, height-field cell edge, friction regime)
                           turns rounding-level differences into visible ones; a kernel defect is not reproduced by such perturbations.
A tight bound on the number of explained envs keeps the explanations from becoming the rule.
"""
import numpy as np

FIELDS = ["root_states", "dof_pos", "dof_vel", "last_dof_targets", "actions", "last_actions", "last_dof_vel", "last_root_vel", "commands",
          "gait_frequency", "gait_process", "filtered_lin_vel", "filtered_ang_vel", "last_feet_pos", "pushing", "episode_length_buf",
          "cmd_resample_time", "delay_steps"]

# relative to max(1, |reference|) per element, worst element per env
STATE_TOL = {"root": 2e-3, "dof_pos": 2e-3, "dof_vel": 1e-2, "torques": 5e-3, "feet_pos": 2e-3, "obs": 5e-3, "priv": 5e-3}
# scaled reward terms (yaml scale x dt, magnitudes 1e-6 .. 1e-2): relative to max(|reference|, REW_FLOOR)
REW_FLOOR = 5e-5          # 1 % of the survival reward of one step (0.25 x 0.02)
REW_TOL = 2e-2
FLAG_TERMS = ("feet_slip", "feet_swing")          # functions of the foot-contact flags
COUNT_TERMS = ("dof_pos_limits", "collision")     # integer counts of threshold crossings


def rel_state(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return (np.abs(a - b) / np.maximum(1.0, np.abs(b))).reshape(a.shape[0], -1).max(axis=1)


def rel_reward(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), REW_FLOOR)


def sync_oracle(env, ref):
    g = {k: env.get_field(k).cpu().numpy() for k in FIELDS}
    n = ref.n
    f = lambda k: g[k].astype(np.float64)
    ref.root, ref.q, ref.qd = f("root_states"), f("dof_pos"), f("dof_vel")
    ref.last_tgt, ref.actions, ref.last_actions = f("last_dof_targets"), f("actions"), f("last_actions")
    ref.last_qd, ref.last_rootvel = f("last_dof_vel"), f("last_root_vel")
    ref.cmd, ref.gait_f, ref.gait_p = f("commands"), f("gait_frequency")[:, 0], f("gait_process")[:, 0]
    ref.filt_lin, ref.filt_ang = f("filtered_lin_vel"), f("filtered_ang_vel")
    ref.last_feet, ref.push = f("last_feet_pos").reshape(n, 2, 3), f("pushing")
    ref.ep_len, ref.cmd_time, ref.delay = (g[k][:, 0].astype(np.int64) for k in ("episode_length_buf", "cmd_resample_time", "delay_steps"))
    ref.step_count = env.common_step_counter


class StepParity:
    def __init__(self, cfg, env, ref, max_explained_frac=0.01, state_tol=None, seed=1234):
        self.cfg, self.env, self.ref = cfg, env, ref
        self.max_explained_frac = max_explained_frac
        self.tol = dict(STATE_TOL)
        self.tol.update(state_tol or {})
        self.rng = np.random.default_rng(seed)
        self.log, self.checked, self.explained, self.flag_flips = [], 0, 0, 0

    # ---- before the step: identical inputs on both sides, and a copy of them for the backward-error probe
    def begin(self):
        sync_oracle(self.env, self.ref)
        r = self.ref
        self.pre = {k: getattr(r, k).copy() for k in ("root", "q", "qd", "last_tgt", "push", "delay")}
        self.pre_cmd_time = r.cmd_time.copy()

    # ---- float64 physics of one env from perturbed inputs: envelope of the post-physics state
    def _envelope(self, e, act, delta, P=48):
        r, cfg, pre = self.ref, self.cfg, self.pre
        rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64)[e:e + 1], P, axis=0).copy()
        root, q, qd, last_tgt = rep(pre["root"]), rep(pre["q"]), rep(pre["qd"]), rep(pre["last_tgt"])
        u = lambda shape: self.rng.uniform(-1.0, 1.0, shape)
        pert = lambda x: x + delta * np.maximum(1.0, np.abs(x)) * u(x.shape)
        root[1:], q[1:], qd[1:] = pert(root[1:]), pert(q[1:]), pert(qd[1:])  # copy 0 = the unperturbed inputs
        root[:, 3:7] /= np.linalg.norm(root[:, 3:7], axis=1, keepdims=True)
        nz = cfg["normalization"]
        a = np.clip(np.asarray(act, dtype=np.float64)[e], -nz["clip_actions"], nz["clip_actions"])
        targets = np.repeat((r.default + cfg["control"]["action_scale"] * a)[None], P, axis=0)
        n = r.n
        com0 = np.array(r.dyn.model.com[0][:]) + r.p["com_off"].reshape(n, 13, 3)[e, 0]
        wrench = pre["push"][e].copy()
        wrench[3:] += np.cross(com0, pre["push"][e, :3])
        r.dyn.substeps_batch(cfg["control"]["decimation"], rep(r.p["mass_scale"]), rep(r.p["com_off"].reshape(n, 39)), rep(r.p["foot_mat"].reshape(n, 6)),
                             rep(r.p["kp"]), rep(r.p["kd"]), rep(r.p["fric"]), r.limits["torque_limits"], root, q, qd, targets, last_tgt,
                             np.repeat(pre["delay"][e:e + 1], P).astype(np.int32), np.repeat(wrench[None], P, axis=0))
        return root, q, qd

    def _inside(self, x, ys, tol):
        lo, hi = ys.min(axis=0), ys.max(axis=0)
        pad = tol * np.maximum(1.0, np.abs(ys[0]))
        return bool(np.all((x >= lo - pad) & (x <= hi + pad)))

    def _explain_physics(self, e, act, kicked):
        """fp32_backward: is the GPU's post-physics state inside the envelope of float64 outcomes for rounding-size input perturbations?"""
        g_root = self.env.root_states[e].cpu().numpy().astype(np.float64)
        g_q = self.env.dof_pos[e].cpu().numpy().astype(np.float64)
        g_qd = self.env.dof_vel[e].cpu().numpy().astype(np.float64)
        for delta in (2e-6, 2e-5):
            root, q, qd = self._envelope(e, act, delta)
            ok = self._inside(g_root[:7], root[:, :7], self.tol["root"]) and self._inside(g_q, q, self.tol["dof_pos"]) and \
                self._inside(g_qd, qd, self.tol["dof_vel"])
            if not kicked:  # a kick adds the same random velocity on both sides after the physics; the envelope has no kick
                ok = ok and self._inside(g_root[7:], root[:, 7:], self.tol["dof_vel"])
            if ok:
                return f"fp32_backward(delta={delta:g})"
        return None

    # ---- after the step
    def check(self, s, act, obs, rew, done, extras, ref_out, kicked=False, teleported=None):
        env, ref, tol = self.env, self.ref, self.tol
        o_ref, p_ref, r_ref, d_ref, t_ref, terms_ref, derived = ref_out
        n = ref.n
        d_gpu = done.cpu().numpy()
        keep = d_gpu == d_ref  # an env whose termination flag flipped (threshold crossing) is reset on one side only: counted by the caller
        if teleported is not None:
            keep = keep & ~teleported
        fc_gpu = env.get_field("feet_contact").cpu().numpy() > 0.5
        fc_ref = np.asarray(derived["feet_contact"]).astype(bool)
        flag_flip = (fc_gpu != fc_ref).any(axis=1)
        err = {"root": rel_state(env.root_states.cpu().numpy(), ref.root), "dof_pos": rel_state(env.dof_pos.cpu().numpy(), ref.q),
               "dof_vel": rel_state(env.dof_vel.cpu().numpy(), ref.qd), "torques": rel_state(env.get_field("torques").cpu().numpy(), derived["torques"]),
               "feet_pos": rel_state(env.get_field("feet_pos").cpu().numpy(), derived["feet_pos"].reshape(n, 6)),
               "obs": rel_state(obs.cpu().numpy(), o_ref), "priv": rel_state(extras["privileged_obs"].cpu().numpy(), p_ref)}
        bad = {k: err[k] > tol[k] for k in err}
        rerr = {"reward": rel_reward(rew.cpu().numpy(), r_ref)}
        for name, v in terms_ref.items():
            rerr[name] = rel_reward(extras["rew_terms"][name].cpu().numpy(), v)
        for k, v in rerr.items():
            bad["rew:" + k] = v > REW_TOL
        any_bad = np.zeros(n, dtype=bool)
        for k, b in bad.items():
            any_bad |= b
        any_bad &= keep
        nk = int(keep.sum())
        lo, hi = ref.limits["dof_pos_limits"][:, 0], ref.limits["dof_pos_limits"][:, 1]
        for e in np.nonzero(any_bad)[0]:
            fields = [k for k, b in bad.items() if b[e]]
            why = []
            state_fields = [k for k in fields if not k.startswith("rew:")]
            rew_fields = [k[4:] for k in fields if k.startswith("rew:")]
            # discrete flags of the final state
            pending = set(rew_fields)
            if flag_flip[e]:
                if pending & (set(FLAG_TERMS) | {"reward"}):
                    why.append("contact_flag")
                pending -= set(FLAG_TERMS) | {"reward"}
            near_limit = bool((np.minimum(np.abs(ref.q[e] - lo), np.abs(ref.q[e] - hi)) < tol["dof_pos"]).any())
            if "dof_pos_limits" in pending and near_limit:
                why.append("limit_crossing")
                pending -= {"dof_pos_limits", "reward"}
            if state_fields or pending:
                w = self._explain_physics(int(e), act, kicked)
                if w is None:
                    worst = {k: float(err[k][e]) for k in state_fields}
                    worst.update({k: float(rerr[k][e]) for k in pending})
                    raise AssertionError(f"step {s} env {e}: outside tolerance and NOT explained: {worst}")
                why.append(w)
            self.log.append((s, int(e), fields, why))
        self.checked += nk
        self.explained += int(any_bad.sum())
        self.flag_flips += int((flag_flip & keep).sum())
        return keep

    def finish(self):
        frac = self.explained / max(self.checked, 1)
        assert frac <= self.max_explained_frac, f"{self.explained} of {self.checked} env-steps needed an explanation ({frac:.4f}): {self.log[:10]}"
        out = {"env_steps": self.checked, "explained": self.explained, "foot_contact_flag_flips": self.flag_flips,
               "by_reason": {w: sum(1 for _, _, _, why in self.log if w in " ".join(why)) for w in ("contact_flag", "limit_crossing", "fp32_backward")}}
        # always in the test output (pytest -s / the captured log of a failure): a regression that starts leaning on the explanations shows here
        print("StepParity:", out, flush=True)
        return out
