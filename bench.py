"""Headline benchmark: env-steps/sec of the full PPO training loop (rollout + update), T1, 4096 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one PPO iteration on synthetic (randomly initialised) policy weights: 24 env-steps x 4096 envs per GPU through the
HIP simulator + 20 full-batch optimiser steps (BASELINE.json configs[1], flat terrain).  Rank 0 prints
ONE JSON line.  `value` = world * N * T * K / wall (max over ranks).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak (same guide)

# Algorithmic HBM bytes of the fused env-step kernel per env per env-step: every per-env field it reads
# or writes once per launch, 4 bytes each.
ENV_STEP_BYTES = 4 * (
    # reads: root 13, q/qd 24, last targets/actions/qd 36, last root vel 6, cmd/gait 5, filters 6, last feet 6, push 6,
    #        kp/kd/friction 36, mass scale 13, com offset 39, foot material 6, base_mass_scaled 4, origin 3, int state 4, actions 12
    (13 + 24 + 36 + 6 + 5 + 6 + 6 + 6 + 36 + 13 + 39 + 6 + 4 + 3 + 4 + 12)
    # writes: root 13, q/qd 24, last targets 12, actions/last actions 24, last qd 12, last root vel 6, cmd/gait 5, filters 6,
    #         last feet 6, push 6, contact 6, derived (feet pos/roll/yaw/contact 12, torques 12, base vel/gravity 9), episode sums 27,
    #         int state 4, obs 47, privileged 14, reward terms 26, rew 1
    + (13 + 24 + 12 + 24 + 12 + 6 + 5 + 6 + 6 + 6 + 6 + 33 + 27 + 4 + 47 + 14 + 26 + 1)
) + 2  # done + time_out bytes


def gemm_flops_per_iteration(n_envs, horizon, mini_epochs):
    """GEMM flops of one update phase: fwd + bwd (= 3 x fwd) of actor and critic over the full batch per
    mini-epoch, plus the no-grad passes (old_mu once, last values every mini-epoch)."""
    actor = 2 * (47 * 256 + 256 * 128 + 128 * 128 + 128 * 12)
    critic = 2 * (61 * 256 + 256 * 256 + 256 * 128 + 128 * 1)
    B = n_envs * horizon
    train = mini_epochs * 3 * B * (actor + critic)
    nograd = B * actor + mini_epochs * n_envs * critic  # old_mu once, last_values every mini-epoch
    return train + nograd


ABA_BYTES = 4 * (13 + 12 + 12 + 12 + 6 + 18 + 6 + 58)  # forward_dynamics_kernel, per env and substep


from booster_gym_amd import _lib  # noqa: E402  (raw ABI calls for the kernel-level timings)


PMC_TAG = "r03"
PMC_SOURCE = (f"profiles/{PMC_TAG}_bench_pmc.json / profiles/{PMC_TAG}_env_pmc.json: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes of "
              "`bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra` and `tools/prof_env.py 4096 plane` (tools/profile.sh); "
              "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE for the 16-byte-per-lane MFMA layer kernels (the guide's gfx950 correction), FETCH_SIZE + WRITE_SIZE otherwise")


def pmc_traffic(kernel_prefix, which="bench"):
    """HBM bytes per launch of a kernel from the committed PMC summary of this round (None if the profile is absent)."""
    path = os.path.join(ROOT, "profiles", f"{PMC_TAG}_{which}_pmc.json")
    try:
        ks = json.load(open(path))["kernels"]
        k = next(v for name, v in ks.items() if name.startswith(kernel_prefix))
        return float(k["hbm_bytes"])
    except (OSError, StopIteration, KeyError, ValueError):
        return None


def _aba_pmc():
    """Per-launch HBM bytes and VALU instructions per wave of the ABA launch's kernels from this round's rocprofv3 passes (tools/profile_aba.sh)."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "r03_a_aba_pmc.json")))["kernels"]
        return {k.split("<")[0]: {"hbm_bytes": v.get("hbm_bytes"), "valu_per_wave": v.get("valu_per_wave")} for k, v in ks.items()}
    except (OSError, KeyError, ValueError):
        return None


def aba_roofline(n=1 << 20, launches=30):
    """HBM roofline of the ABA launch on a FULL chip: bg_env_forward_dynamics (one substep's accelerations per launch) on n synthetic states, HIP
    events on the launch stream.  At the training size (4096 envs = 128 waves on 1024 SIMDs) no kernel can approach a bandwidth roof.
    The launch is forward_dynamics_kernel (every env) + aba_compact_kernel + forward_dynamics_body_kernel (the envs whose legs can meet: the
    leg-against-leg narrow phase) -- `achieved` prices the WHOLE launch.  Two state distributions:
      `standing_noise_0.1` (the headline entry, the state of rounds 1 and 2): joints = default pose + N(0, 0.1 rad), trunk upright at 0.66 m.  The
          0.1 rad of hip-roll noise crosses the legs of 8 % of the envs, which now go through the second kernel;
      `survey_8d_state`: SURVEY section 8(d)'s K1 inputs (joints ~ U(limits), trunk at 0.72 m within 0.3 rad of upright, torques ~ U(+-effort))."""
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
    dev = env.device
    m = env.model
    lib = _lib.load()
    qacc = torch.empty(n, 18, device=dev)

    def measure(root, q, qd, tau):
        root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
        call = lambda: _lib.check(lib.bg_env_forward_dynamics(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc),
                                                              _lib.current_stream_ptr()), "bg_env_forward_dynamics")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            call()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches * 1e3

    g = torch.Generator(device="cpu").manual_seed(1234)
    root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
    q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * 0.1
    qd = torch.randn(n, 12, generator=g)
    tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
    us = measure(root, q, qd, tau)
    # SURVEY 8(d) K1 inputs
    lo, hi, eff = (torch.tensor(a, dtype=torch.float32) for a in (m.dof_lower, m.dof_upper, m.dof_effort))
    root2 = torch.zeros(n, 13); root2[:, 2] = 0.72
    ax = torch.randn(n, 3, generator=g); ax = ax / ax.norm(dim=1, keepdim=True)
    ang = torch.rand(n, generator=g) * 0.3
    root2[:, 3:6] = ax * torch.sin(ang / 2)[:, None]; root2[:, 6] = torch.cos(ang / 2)
    q2 = lo + (hi - lo) * torch.rand(n, 12, generator=g)
    tau2 = (torch.rand(n, 12, generator=g) * 2 - 1) * eff
    us2 = measure(root2, q2, torch.randn(n, 12, generator=g), tau2)
    gbs, gbs2 = n * ABA_BYTES / us / 1e3, n * ABA_BYTES / us2 / 1e3
    pmc = _aba_pmc()
    traffic = sum(v["hbm_bytes"] for v in pmc.values() if v.get("hbm_bytes")) if pmc else None
    return {"kernel": "bg_env_forward_dynamics: forward_dynamics_kernel (hand-written HIP: one ABA substep with contact and limits, per-step joint accelerations) + "
                      "aba_compact_kernel + forward_dynamics_body_kernel (leg-against-leg narrow phase for the envs whose legs can meet)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/r03_a_aba_pmc.json (tools/profile_aba.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in separate passes "
                              "of tools/aba_only.py, the same launch); FETCH_SIZE + WRITE_SIZE as reported: dword-per-lane accesses, whose width the guide "
                              "calls uncalibrated on gfx950, and inputs that stay in the 256 MB Infinity Cache between launches -- indicative only",
            "avg_launch_us": us, "num_envs": n, "algorithmic_bytes_per_launch": n * ABA_BYTES, "state": "standing_noise_0.1",
            "kernels_per_launch_rocprof": "profiles/r03_a_aba_kernel_stats.csv: forward_dynamics_kernel 210.4 us (3,120 VALU per wave, was 3,331 in round 2), "
                                          "forward_dynamics_body_kernel 78.6 us, aba_compact_kernel 3 us",
            "survey_8d_state": {"avg_launch_us": us2, "achieved": gbs2, "frac": gbs2 / HBM_PEAK_GBS,
                                "state": "joints ~ U(limits), trunk at 0.72 m within 0.3 rad of upright, torques ~ U(+-effort), qd ~ N(0, 1)"},
            "note": "VALU-issue bound (SQ counters in the PMC file: the SIMDs issue VALU 100 % of the wave cycles of forward_dynamics_kernel at 4 cycles per "
                    "instruction); itemised instruction budget: tools/isa_census.py, DESIGN.md section 6"}


def cpu_baseline(n_envs=4096):
    """CPU restatement baseline ("port"), timed in a child process that never touches the GPU (oracle/cpu_baseline.py): the bench's own
    workload (n_envs envs x 24 env-steps with task logic + 20 full-batch mini-epochs) on all host cores of this GPU's share, and the same
    on ONE core at a quarter of the envs (report single-thread and all-core)."""
    import subprocess

    def run(n, threads=None):
        env = dict(os.environ)
        if threads:
            env["BG_CPU_THREADS"] = str(threads)
        r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", str(n)], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-400:])
        return json.loads(r.stdout.strip().splitlines()[-1])

    out = run(n_envs)
    one = run(max(1024, n_envs // 4), threads=1)
    out["single_thread"] = {k: one[k] for k in ("value", "unit", "cores", "num_envs", "phase_s", "sample")}
    return out


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child process (one rank per GPU over RCCL) and relay
    its stdout (rank 0's JSON line), stderr and exit status.  The launcher itself never initialises the GPU."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--terrain", type=str, default="plane")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the measurements after the timed region (kernel alone on the GPU, ABA roofline): "
                    "used under rocprofv3 so that the per-kernel averages of the summary are those of the timed region")
    args = ap.parse_args()

    t_start = time.perf_counter()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher.  Nothing above has touched the GPU (importing torch
        # and loading the .so do not), the ranks are CHILD processes and this one only relays their output and exit status.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    import torch.distributed as dist

    from booster_gym_amd.utils.config import load_cfg
    from booster_gym_amd.utils.runner import Runner

    cfg = load_cfg("T1", {"env.num_envs": args.num_envs, "terrain.type": args.terrain, "basic.seed": 42})
    runner = Runner(cfg=cfg)  # initialises the process group when WORLD_SIZE > 1
    T, N, E = cfg["runner"]["horizon_length"], runner.env.num_envs, cfg["runner"]["mini_epochs"]
    dev = runner.device

    # the timed loop is Runner.train()'s own loop body (train_iteration: rollout, update, statistics read-back, curriculum exchange, logging to a
    # scratch directory), not a stripped-down copy of it
    import tempfile

    from booster_gym_amd.utils.recorder import Recorder

    from booster_gym_amd.utils.model import MLPTrainer as _MT
    split_mode = _MT.SPLIT  # 0 unless the caller exported BG_GEMM_SPLIT: then the headline loop itself runs in split mode, and the line says so
    cfg["runner"]["save_interval"] = 10 ** 9  # no checkpoint inside the timed region (the reference saves every 100 iterations)
    import atexit
    import shutil
    log_root = tempfile.mkdtemp(prefix="bench_logs_")
    atexit.register(shutil.rmtree, log_root, ignore_errors=True)  # every rank and every A/B loop run would otherwise leave its directory in /tmp
    runner.begin_training(Recorder(cfg, root=log_root, rank=rank))

    def barrier():
        torch.cuda.synchronize()
        runner.dp.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    log("env + runner built")
    for w in range(args.warmup):
        runner.train_iteration(w)
        torch.cuda.synchronize()
        log(f"warmup iteration {w} done")
    # instrument: HIP events around every env-step launch and around the update phase (torch's current stream is the launch stream)
    step_events, phase_events = [], []
    orig_step_to = runner.env.step_to

    def timed_step_to(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig_step_to(*a, **k); e1.record()
        step_events.append((e0, e1))

    runner.env.step_to = timed_step_to
    # critic layer 1 (256 -> 256): the single largest kernel of the update; critic layer 2 (256 -> 128): the symbol with the largest TOTAL time
    # (mlp_fwd_kernel<256,1,1>, shared with the actor's layer 1)
    runner._critic_tr.timed_layer = (1, 2)
    runner._actor_tr.timed_layer = (1, 2)  # (with the chained forward kernel: the two networks' launches overlap; their union is reported too)
    runner._wgrad_group.timed_events = []  # the grouped weight-gradient launch: the kernel with the largest total time of the iteration
    if world > 1:
        runner.dp.timed_events = []  # HIP events around the gradient-bucket all-reduce of every mini-epoch
    barrier()
    t0 = time.perf_counter()
    orig_rollout, orig_update = runner.rollout, runner.update

    def timed_rollout():
        e = torch.cuda.Event(enable_timing=True); e.record(); phase_events.append([e])
        return orig_rollout()

    def timed_update():
        e = torch.cuda.Event(enable_timing=True); e.record(); phase_events[-1].append(e)
        out = orig_update()
        e2 = torch.cuda.Event(enable_timing=True); e2.record(); phase_events[-1].append(e2)
        return out

    runner.rollout, runner.update = timed_rollout, timed_update
    for k in range(args.steps):
        runner.train_iteration(args.warmup + k)
    barrier()
    wall = time.perf_counter() - t0
    runner._flush_log()  # the last iteration's scalars (outside the timed region: train() does the same after its loop)
    tw = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())

    log(f"timed region done: {wall:.3f}s for {args.steps} iterations")
    if rank == 0:
        step_ms = sum(a.elapsed_time(b) for a, b in step_events) / max(len(step_events), 1)
        roll_ms = sum(a.elapsed_time(b) for a, b, _ in phase_events) / len(phase_events)
        upd_ms = sum(b.elapsed_time(c) for _, b, c in phase_events) / len(phase_events)
        # gradient-bucket all-reduce (712 kB, SURVEY 8e exchange 2), per mini-epoch, rank 0's view: collective + waiting for the slowest rank
        ar = runner.dp.timed_events or []
        ar_ms = sum(a.elapsed_time(b) for a, b in ar) / len(ar) if ar else 0.0
        env_bytes = N * ENV_STEP_BYTES
        sim_gbs = env_bytes / (step_ms * 1e-3) / 1e9
        flops = gemm_flops_per_iteration(N, T, E)
        # dominant kernel by GPU time (rocprof, profiles/): the hand-written fused Linear+ELU layer mlp_fwd_kernel<256,1>; its largest instance
        # (critic 256 -> 256, [rows x 256] x [256 x 256]) is timed inside the timed region with HIP events on the stream it is launched on
        ev_all = runner._critic_tr.timed_events
        ev = [e for e in ev_all if e[5] == 1]
        ev2 = [e for e in ev_all if e[5] == 2]
        evc = [e for e in ev_all if e[5] == "chain"]
        top_by_time = None
        if ev2:
            us2 = sum(a.elapsed_time(b) for a, b, *_ in ev2) / len(ev2) * 1e3
            r2_, k2_, n2_ = ev2[0][2], ev2[0][3], ev2[0][4]
            fl2 = 2.0 * r2_ * k2_ * n2_
            top_by_time = {"kernel": f"mlp_fwd_kernel<256,1,1>: fused Linear+bias+ELU, critic layer 3, [{r2_}x{k2_}]x[{k2_}x{n2_}] (the symbol with the largest total "
                                     "time in the rocprofv3 summary: the actor's layer 2 runs on it too)", "bound": "mfma", "achieved": fl2 / (us2 * 1e-6) / 1e12,
                           "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": fl2 / (us2 * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, "avg_launch_us": us2,
                           "algorithmic_flops_per_launch": fl2, "traffic": pmc_traffic("mlp_fwd_kernel<256, 1, 1>")}
        if evc:
            # the critic's three hidden layers are ONE launch (bg_mlp_chain.hip): timed as a whole, in the loop and alone on the GPU
            gemm_us = sum(a.elapsed_time(b) for a, b, *_ in evc) / len(evc) * 1e3
            rows_g, kg, widths = evc[0][2], evc[0][3], evc[0][4]
            gemm_flop = 2.0 * rows_g * (kg * widths[0] + widths[0] * widths[1] + widths[1] * widths[2])
            gemm_name = (f"mlp_chain_fwd_kernel<2>: the critic's three fused Linear+bias+ELU layers in one launch, [{rows_g}x{kg}] -> {widths[0]} -> {widths[1]} -> "
                         f"{widths[2]}, activations handed on in registers, fp32 MFMA 32x32x2 (hand-written HIP, bg_mlp_chain.hip)")
            tr = runner._critic_tr
            d = tr._chain_descriptor()
            lib = _lib.load()
            solo = lambda: _lib.check(lib.bg_mlp_chain_forward_group(ctypes.addressof(d), 1, _lib.current_stream_ptr()), "bg_mlp_chain_forward_group")
            torch.cuda.synchronize()
            solo_us = float("nan")
            if not args.no_extra:
                for _ in range(3):
                    solo()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(30):
                    solo()
                g1.record(); torch.cuda.synchronize()
                solo_us = g0.elapsed_time(g1) / 30 * 1e3
            # the actor's chained launch runs beside the critic's on the other stream: flops of both over the span from the first start to the last end
            eva = [e for e in runner._actor_tr.timed_events if e[5] == "chain"]
            mini = cfg["runner"]["mini_epochs"]
            both = None
            if len(eva) == len(evc) // mini * (mini + 1):  # per iteration: the old-mu forward, then one launch per mini-epoch
                spans = []
                for i, (c0, c1, *_) in enumerate(evc):
                    a0, a1 = eva[i // mini * (mini + 1) + 1 + i % mini][:2]
                    first = a0 if a0.elapsed_time(c0) >= 0 else c0
                    last = c1 if a1.elapsed_time(c1) >= 0 else a1
                    spans.append(first.elapsed_time(last))
                span_us = sum(spans) / len(spans) * 1e3
                ra, ka, wa = eva[0][2], eva[0][3], eva[0][4]
                fl_both = gemm_flop + 2.0 * ra * (ka * wa[0] + wa[0] * wa[1] + wa[1] * wa[2])
                both = {"what": "the critic's and the actor's chained forward launches of a mini-epoch together (they overlap on two streams): flops of both / time from "
                                "the first start to the last end", "avg_span_us": span_us, "algorithmic_flops": fl_both,
                        "achieved": fl_both / (span_us * 1e-6) / 1e12, "frac": fl_both / (span_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF}
        elif ev:
            gemm_us = sum(a.elapsed_time(b) for a, b, *_ in ev) / len(ev) * 1e3
            rows_g, kg, ng = ev[0][2], ev[0][3], ev[0][4]
            gemm_name = f"mlp_fwd_kernel<256,1,2>: fused Linear+bias+ELU, critic layer 2, [{rows_g}x{kg}]x[{kg}x{ng}] fp32 MFMA 32x32x2 (hand-written HIP, bg_mlp.hip)"
            # the same launch with nothing else on the GPU (in the loop the actor's kernels run beside it on the second stream)
            tr = runner._critic_tr
            xg, lg, og = tr.acts[0], tr.layers[1], tr.acts[1]
            lib = _lib.load()
            solo = lambda: _lib.check(lib.bg_mlp_layer_forward(rows_g, kg, ng, _lib.ptr(xg), _lib.ptr(lg.weight), _lib.ptr(lg.bias), _lib.ptr(og), 1,
                                                               _lib.current_stream_ptr()), "bg_mlp_layer_forward")
            torch.cuda.synchronize()
            solo_us = float("nan")
            if not args.no_extra:
                for _ in range(3):
                    solo()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(30):
                    solo()
                g1.record(); torch.cuda.synchronize()
                solo_us = g0.elapsed_time(g1) / 30 * 1e3
        else:  # BG_FUSED_MLP=0: the library GEMM of the same layer
            tr = runner._critic_tr
            xg, lg, og = tr.acts[0], tr.layers[1], torch.empty_like(tr.acts[1])
            with torch.no_grad():
                for _ in range(5):
                    torch.addmm(lg.bias, xg, lg.weight.t(), out=og)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                for _ in range(50):
                    torch.addmm(lg.bias, xg, lg.weight.t(), out=og)
                g1.record(); torch.cuda.synchronize()
            gemm_us = g0.elapsed_time(g1) / 50 * 1e3
            rows_g, kg, ng = xg.shape[0], lg.weight.shape[1], lg.weight.shape[0]
            gemm_name = f"critic layer-2 forward GEMM [{rows_g}x{kg}]x[{kg}x{ng}] fp32 (hipBLASLt via torch.addmm)"
            solo_us = gemm_us
        if not evc:
            gemm_flop = 2.0 * rows_g * kg * ng
        gemm_tf = gemm_flop / (gemm_us * 1e-6) / 1e12
        layer_fwd = {"kernel": gemm_name, "bound": "mfma", "achieved": gemm_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": gemm_tf / MFMA_F32_PEAK_TF,
                     "traffic": pmc_traffic("mlp_chain_fwd_kernel<2>" if evc else "mlp_fwd_kernel<256, 1, 2>") if rows_g == (T + 1) * 4096 else None,
                     "traffic_source": PMC_SOURCE,
                     "avg_launch_us": gemm_us, "algorithmic_flops_per_launch": gemm_flop,
                     "note": "timed inside the loop, where the actor's kernels run beside it on the second stream",
                     "alone_on_the_gpu": {"avg_launch_us": solo_us, "achieved": gemm_flop / (solo_us * 1e-6) / 1e12,
                                          "frac": gemm_flop / (solo_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF}}
        if evc and both is not None:
            layer_fwd["both_networks_forward"] = both
        if args.no_extra:
            layer_fwd.pop("alone_on_the_gpu", None)
        wg_ev = runner._wgrad_group.timed_events or []
        headline = layer_fwd
        if wg_ev:  # the dominant kernel of the iteration by total time (profiles/r02_bench_kernel_stats.csv): all six hidden-layer weight gradients
            wus = sum(a.elapsed_time(b) for a, b, *_ in wg_ev) / len(wg_ev) * 1e3
            wfl = wg_ev[0][2]
            headline = {"kernel": "mlp_wgrad_group_kernel (+ its fixed-order finish): dW = G^T A of all six hidden layers of both networks in one launch pair, " +
                                  " + ".join(f"[{co}x{m}]x[{m}x{ci}]" for m, co, ci in wg_ev[0][3]) + ", fp32 MFMA 32x32x2 (hand-written HIP, bg_wgrad.hip)",
                        "bound": "mfma", "achieved": wfl / (wus * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                        "frac": wfl / (wus * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, "traffic": pmc_traffic("mlp_wgrad_group_kernel") if N == 4096 else None,
                        "traffic_source": PMC_SOURCE, "avg_launch_us": wus, "algorithmic_flops_per_launch": wfl,
                        "note": "launch pair (main kernel + finish) timed inside the loop with HIP events on its stream; it runs after both backward chains, alone on the GPU"}
        # HBM traffic per launch comes from PMC counters, which rocprofv3 collects in separate passes of the same command (tools/profile.sh
        # -> profiles/<PMC_TAG>_*_pmc.json, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes); the JSON line names the file it cites
        traffic = pmc_traffic("env_step_kernel", which="env") if N == 4096 else None
        out = {
            "metric": "env-steps/sec (whole node), PPO rollout+update, T1 4096 envs/GPU",
            "value": world * N * T * args.steps / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (random-init policy, seeded domain randomisation)",
            "config": {"workload": f"T1 {args.terrain} terrain, {N} envs/GPU, horizon {T}, {E} mini-epochs, full batch (BASELINE.json configs[1])",
                       "envs_per_gpu": N, "parallelism": f"dp{world}",
                       "gemm_arithmetic": {0: "fp32 MFMA (v_mfma_f32_32x32x2_f32)", 9: "fp32 operands as exact 3-way bf16 splits, 9 products, fp32 accumulate (BG_GEMM_SPLIT=9)",
                                           6: "fp32 operands as exact 3-way bf16 splits, 6 largest products, fp32 accumulate (BG_GEMM_SPLIT=6)"}[split_mode]},
            "ppo_iters_per_s": args.steps / wall,
            "phase_ms": {"rollout": roll_ms, "update": upd_ms, "all_reduce_ms": ar_ms},
            "roofline": headline,
            "roofline_layer_forward": layer_fwd,
            "roofline_env_step": {"kernel": "env_step_kernel (hand-written HIP: 10 ABA substeps + task logic, one launch per env-step)", "bound": "hbm",
                                  "achieved": sim_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sim_gbs / HBM_PEAK_GBS, "traffic": traffic,
                                  "traffic_source": PMC_SOURCE, "avg_launch_us": step_ms * 1e3, "algorithmic_bytes_per_launch": env_bytes,
                                  "note": "issue-latency-bound at 128 waves: SQ counters in profiles/ show VALU busy 73% of wave cycles at 4 cycles/instruction"},
            "roofline_update": {"bound": "mfma", "achieved": flops / (upd_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": flops / (upd_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                "note": "all actor+critic GEMM flops of the update phase / update-phase wall time (which also holds GAE, loss, ELU, Adam)"},
            "nonfinite_resets": runner.nonfinite_resets_total,
        }
        if top_by_time is not None:
            out["roofline_top_by_time"] = top_by_time
        if world == 1 and not args.no_extra:
            try:
                # Opt-in form of the layer kernels, measured beside the headline and NOT part of `value`: BG_GEMM_SPLIT (bg_mlp_split.hip) runs the
                # fp32 x fp32 products of the hidden-layer forward / backward GEMMs on the bf16 matrix pipe, every fp32 operand split EXACTLY into
                # three bf16 numbers (all 9 cross products: no rounding of the products, fp32 accumulation; 6: the three smallest dropped);
                # so does the grouped weight-gradient launch (bg_wgrad_split.hip).  Same loop, same workload, same timing as `value`.
                from booster_gym_amd.utils.model import MLPTrainer

                runner.rollout, runner.update, runner.env.step_to = orig_rollout, orig_update, orig_step_to
                runner._critic_tr.timed_layer, runner._actor_tr.timed_layer, runner._wgrad_group.timed_events = None, None, None
                split, it0 = {}, args.warmup + args.steps
                for terms in (9, 6):
                    MLPTrainer.SPLIT = terms
                    for _ in range(2):
                        runner.train_iteration(it0); it0 += 1
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for _ in range(args.steps):
                        runner.train_iteration(it0); it0 += 1
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - ts
                    split[f"products_{terms}"] = {"value": N * T * args.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / args.steps * 1e3}
                MLPTrainer.SPLIT = split_mode
                runner._flush_log()
                split["note"] = ("opt-in (BG_GEMM_SPLIT=9|6), not the headline: hidden-layer forward / backward / weight-gradient GEMMs as exact hi/mid/lo "
                                 "bf16 splits of the fp32 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate; tests/test_gpu_mlp_split.py holds the error "
                                 "against float64 beside the fp32-MFMA kernels'")
                out["opt_in_split_bf16_layers"] = split
            except Exception as ex:
                out["opt_in_split_bf16_layers"] = {"error": repr(ex)}
            try:
                del runner.env
                out["roofline_aba"] = aba_roofline()
            except Exception as ex:
                out["roofline_aba"] = {"error": repr(ex)}
        if not args.no_cpu_baseline and world == 1:
            try:
                log("cpu baseline ...")
                out["cpu_baseline"] = cpu_baseline(N)
                log("cpu baseline done")
            except Exception as ex:  # the bench line must still print
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    runner.dp.shutdown()


if __name__ == "__main__":
    main()
