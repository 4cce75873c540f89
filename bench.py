"""Headline benchmark: env-steps/sec of the full PPO training loop (rollout + update), T1, 4096 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one PPO iteration on synthetic (randomly initialised) policy weights: 24 env-steps x 4096 envs per GPU through the
HIP simulator + 20 full-batch optimiser steps (BASELINE.json configs[1], flat terrain).  Rank 0 prints
ONE JSON line.  `value` = world * N * T * K / wall (max over ranks).

`roofline` = both networks' chained forward pass, one launch (bg_mlp_chain_split.hip: fp32 operands as exact three-way bf16 splits, all 9 products on the
bf16 matrix pipe, fp32 accumulation), priced against the 9-product equivalent of the bf16 pipe's peak AND against the fp32 matrix pipe's;
`roofline_backward` = both backward-data chains, one launch (bg_mlp_chain_split_bwd.hip); `roofline_wgrad` = the grouped weight gradients (bg_wgrad_split.hip:
the same arithmetic; the largest single launch of the iteration, profiles/r06_bench_kernel_stats.csv); `fp32_mfma_loop` = the SAME loop in the same run
with every GEMM on the fp32 matrix pipe (last round's headline path), `fp32_mfma_weight_gradients_loop` / `split_forward_only_loop` = the steps between;
`gemm_errors_vs_float64` = the measured error pair (split kernel, fp32-MFMA kernel) of each GEMM family; `roofline_env_step`, `roofline_aba` = the simulator kernels against the HBM roof; `other_configs` = BASELINE
configs[2] and [4] through the same loop; `cpu_baseline` = the oracle's CPU restatement of the same workload on the host cores.
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
MFMA_BF16_PEAK_TF = 2516.6 # bf16 MFMA dense peak (same guide: 16 x the fp32-input rate)
SPLIT9_PEAK_TF = MFMA_BF16_PEAK_TF / 9.0  # fp32-exact flops per second of the bf16 pipe when every fp32 x fp32 product costs 9 bf16 x bf16 products

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


PMC_TAG = "r06"
KERNEL_EVENTS_EVERY = 4  # the timed region's iterations whose kernels are bracketed by HIP timing events (see arm_kernel_events)
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


ABA_PMC_FILE = f"profiles/{PMC_TAG}_aba_pmc.json"            # tools/profile_aba.sh <tag>: the one file both the numbers and their citation come from
ABA_STATS_FILE = f"profiles/{PMC_TAG}_aba_kernel_stats.csv"
ABA_BYTES_SURVEY_8D = 500  # SURVEY section 8(d)'s figure for the same kernel: it counts 13 + 39 inertial parameters where this build reads 58 (+ foot materials) and writes the 6 foot-force floats


def pmc_ratio(kernel_prefix, num, den, which="env"):
    """Ratio of two counters of a kernel in the committed PMC summary (e.g. SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = the share of its waves' cycles in which
    the vector ALU is busy); None if the profile is absent."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", f"{PMC_TAG}_{which}_pmc.json")))["kernels"]
        k = next(v for name, v in ks.items() if name.startswith(kernel_prefix))
        return float(k[num]["mean"]) / float(k[den]["mean"])  # (both in quad-cycles)
    except (OSError, StopIteration, KeyError, ValueError, ZeroDivisionError):
        return None


def _aba_pmc():
    """Per-launch HBM bytes and VALU instructions per wave of the ABA launch's kernels from this round's rocprofv3 passes (tools/profile_aba.sh)."""
    try:
        ks = {k: v for k, v in json.load(open(os.path.join(ROOT, ABA_PMC_FILE)))["kernels"].items() if "pk_kernel" not in k}
        return {k.split("<")[0]: {"hbm_bytes": v.get("hbm_bytes"), "valu_per_wave": v.get("valu_per_wave")} for k, v in ks.items()}
    except (OSError, KeyError, ValueError):
        return None


def aba_roofline(n=1 << 20, launches=50):
    """HBM roofline of the ABA launch on a FULL chip: bg_env_forward_dynamics (one substep's accelerations per launch) on n synthetic states, HIP
    events on the launch stream.  At the training size (4096 envs = 128 waves on 1024 SIMDs) no kernel can approach a bandwidth roof.
    The launch is ONE kernel, forward_dynamics_kernel; the envs whose legs can meet get the leg-against-leg narrow phase inside it, item-parallel
    through LDS (round 3: a second kernel).  Two state distributions:
      `standing_noise_0.1` (the headline entry, the state of rounds 1 to 3): joints = default pose + N(0, 0.1 rad), trunk upright at 0.66 m.  The
          0.1 rad of hip-roll noise brings the legs of 9 % of the envs close to each other (93 % of the 32-env waves have at least one);
      `survey_8d_state`: SURVEY section 8(d)'s K1 inputs (joints ~ U(limits), trunk at 0.72 m within 0.3 rad of upright, torques ~ U(+-effort))."""
    from booster_gym_amd.envs import T1
    from booster_gym_amd.utils.config import load_cfg

    env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
    dev = env.device
    m = env.model
    lib = _lib.load()
    qacc = torch.empty(n, 18, device=dev)

    cold = []

    def measure(root, q, qd, tau, entry="bg_env_forward_dynamics"):
        root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
        fn = getattr(lib, entry)
        call = lambda: _lib.check(fn(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc), _lib.current_stream_ptr()), entry)
        # The part's clock under this kernel is a transient for the first ~40 ms of back-to-back launches (tools/aba_series.py, profiles/r04_aba_series.txt:
        # 226 us for launches 1-4 from idle, up to 270 around launch 12, then a steady decline to 205-206 us from launch ~160 on, where it stays).
        # The roofline figure is the sustained rate: 200 untimed launches, then the median of three series.
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            call()
        c0.record()
        for _ in range(30):
            call()
        c1.record()
        for _ in range(170):
            call()
        torch.cuda.synchronize()
        cold.append(c0.elapsed_time(c1) / 30 * 1e3)  # launches 4-33 from idle: what rounds 1-3 reported
        reps = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                call()
            e1.record(); torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) / launches * 1e3)
        return sorted(reps)[1]  # the median of three back-to-back series

    g = torch.Generator(device="cpu").manual_seed(1234)
    root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
    q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * 0.1
    qd = torch.randn(n, 12, generator=g)
    tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
    us = measure(root, q, qd, tau)
    us_pk = measure(root, q, qd, tau, entry="bg_env_forward_dynamics_packed")
    # SURVEY 8(d) K1 inputs
    lo, hi, eff = (torch.tensor(a, dtype=torch.float32) for a in (m.dof_lower, m.dof_upper, m.dof_effort))
    root2 = torch.zeros(n, 13); root2[:, 2] = 0.72
    ax = torch.randn(n, 3, generator=g); ax = ax / ax.norm(dim=1, keepdim=True)
    ang = torch.rand(n, generator=g) * 0.3
    root2[:, 3:6] = ax * torch.sin(ang / 2)[:, None]; root2[:, 6] = torch.cos(ang / 2)
    q2 = lo + (hi - lo) * torch.rand(n, 12, generator=g)
    tau2 = (torch.rand(n, 12, generator=g) * 2 - 1) * eff
    qd2 = torch.randn(n, 12, generator=g)
    us2 = measure(root2, q2, qd2, tau2)
    us2_pk = measure(root2, q2, qd2, tau2, entry="bg_env_forward_dynamics_packed")
    gbs, gbs2 = n * ABA_BYTES / us / 1e3, n * ABA_BYTES / us2 / 1e3
    valu_busy = pmc_ratio("forward_dynamics_kernel", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", which="aba")
    valu_busy_pk = pmc_ratio("forward_dynamics_pk_kernel", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", which="aba")
    pmc = _aba_pmc()
    traffic = sum(v["hbm_bytes"] for v in pmc.values() if v.get("hbm_bytes")) if pmc else None
    return {"kernel": "bg_env_forward_dynamics = forward_dynamics_kernel (hand-written HIP, ONE launch: an ABA substep with sole contact, joint limits and the "
                      "leg-against-leg narrow phase item-parallel through LDS; per-step joint accelerations)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": f"{ABA_PMC_FILE} (tools/profile_aba.sh {PMC_TAG}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in separate passes "
                              "of tools/aba_only.py, the same launch); FETCH_SIZE + WRITE_SIZE as reported: dword-per-lane accesses, whose width the guide "
                              "calls uncalibrated on gfx950 -- indicative only",
            "traffic_note": "inputs partly Infinity-Cache-resident: identical back-to-back launches keep about half of the 575 MB of inputs in the 256 MB cache, so "
                            "`traffic` is below the algorithmic bytes and `frac` is an algorithmic-bytes rate against the HBM roof, not measured HBM traffic",
            "bytes_per_env": {"this_build": ABA_BYTES, "survey_8d": ABA_BYTES_SURVEY_8D},
            "frac_with_survey_8d_bytes": n * ABA_BYTES_SURVEY_8D / us / 1e3 / HBM_PEAK_GBS,
            "avg_launch_us": us, "num_envs": n, "algorithmic_bytes_per_launch": n * ABA_BYTES, "state": "standing_noise_0.1",
            "timing": "sustained rate: 200 untimed launches, then the median of three series of 50 (HIP events on the launch stream).  launches_4_to_33_from_idle_us is "
                      "the figure rounds 1-3 reported as avg_launch_us (round 3: 305 us): the clock under this kernel is a transient for the first ~40 ms "
                      "(profiles/r04_aba_series.txt)",
            "launches_4_to_33_from_idle_us": cold[0],
            "kernels_per_launch_rocprof": f"{ABA_STATS_FILE}: forward_dynamics_kernel over 1,200 launches (the first ~160 in the clock transient); "
                                          f"PMC passes (12 launches each): {ABA_PMC_FILE}",
            "survey_8d_state": {"avg_launch_us": us2, "achieved": gbs2, "frac": gbs2 / HBM_PEAK_GBS,
                                "frac_with_survey_8d_bytes": n * ABA_BYTES_SURVEY_8D / us2 / 1e3 / HBM_PEAK_GBS, "launches_4_to_33_from_idle_us": cold[2],
                                "state": "joints ~ U(limits), trunk at 0.72 m within 0.3 rad of upright, torques ~ U(+-effort), qd ~ N(0, 1)"},
            "packed_form": {"kernel": "bg_env_forward_dynamics_packed = forward_dynamics_pk_kernel: one env per lane, both legs in 64-bit register pairs "
                                      "(v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), ~420 registers, one wave per SIMD",
                            "avg_launch_us": us_pk, "frac": n * ABA_BYTES / us_pk / 1e3 / HBM_PEAK_GBS, "survey_8d_state_avg_launch_us": us2_pk,
                            "valu_busy_share_of_wave_cycles": valu_busy_pk,
                            "note": "43 % fewer VALU instructions per env than the lane-per-leg kernel and no faster: a lone wave per SIMD keeps the vector ALU "
                                    "busy about half of its life (HISTORY.md, round 5); not the default"},
            "valu_busy_share_of_wave_cycles": valu_busy, "waves_per_simd": 2,
            "note": "vector-ALU bound: two waves share a SIMD and each has the vector ALU busy in valu_busy_share_of_wave_cycles of its cycles (SQ_ACTIVE_INST_VALU / "
                    "SQ_WAVE_CYCLES of forward_dynamics_kernel in the PMC file), i.e. the SIMD's ALU is busy twice that share; itemised instruction budget: tools/isa_census.py, profiles/r04_aba_isa_census_plane_t1.json, DESIGN.md section 6"}


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
    # `cores` = what this process may really use (affinity mask and cgroup CPU quota: oracle/cpu_baseline.py).  When the host shows more cores than that,
    # the same workload once more with one thread per VISIBLE core, so that the line carries SURVEY 8(d)'s "OMP_NUM_THREADS = nproc" figure beside it
    if out.get("host_cores_visible", 0) > out["cores"]:
        try:
            allc = run(n_envs, threads=out["host_cores_visible"])
            out["all_visible_cores"] = {k: allc[k] for k in ("value", "unit", "cores", "num_envs", "phase_s")}
        except Exception as ex:
            out["all_visible_cores"] = {"error": repr(ex)}
    return out


def gemm_error_pairs(rows=16384):
    """Measured error against float64 of each GEMM family of the update, the split-bf16 chain beside the fp32-MFMA kernels of the same op on the same
    inputs (the weights and the layer shapes of the two networks, synthetic inputs of the update's scale): rms, largest, and mean signed error."""
    import ctypes as C

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_mlp_chain_split as TF
    import test_gpu_mlp_chain_split_bwd as TB

    lib, st = _lib.load(), _lib.current_stream_ptr()

    def stat(y, ref):
        e = y.double() - ref
        return {"rms": float(e.pow(2).mean().sqrt()), "max": float(e.abs().max()), "mean_signed": float(e.mean())}

    out = {}
    for name, dims, kr in (("critic", (64, 256, 256, 128), 61), ("actor", (64, 256, 128, 128), 47)):
        d, x, Ws, bs, ys, Ps = TF._case(rows, dims, seed=7, k_real=kr)
        _lib.check(lib.bg_mlp_chain_forward_split(C.addressof(d), 1, st), "bg_mlp_chain_forward_split")
        zs = TF._fp32_chain(rows, dims, x, Ws, bs)
        ref = x.double()[:, :kr]
        for l in range(3):
            ref = torch.nn.functional.elu(ref @ Ws[l].double().t() + bs[l].double())
            out[f"forward_{name}_layer{l + 1}"] = {"split9_chain": stat(ys[l][:rows], ref), "fp32_mfma_chain": stat(zs[l][:rows], ref)}
        bd = (dims[1], dims[2], dims[3])
        db, t = TB._case(rows, bd, 7, 0)
        fin = _lib.ReduceProblem()
        _lib.check(lib.bg_mlp_chain_backward_split(C.addressof(db), 1, fin, st), "bg_mlp_chain_backward_split")
        from booster_gym_amd.utils.utils import reduce_group
        reduce_group([fin])
        r2 = (t["G3"].double() @ t["W3"].double()) * TB._elup(t["A2"][:rows].double())
        r1 = (r2 @ t["W2"].double()) * TB._elup(t["A1"][:rows].double())
        f32 = TB._fp32_layers(rows, t)
        for nm, y, ref, (z, zb), b in (("G2", t["G2"], r2, f32[0], t["b2"]), ("G1", t["G1"], r1, f32[1], t["b1"])):
            out[f"backward_{name}_{nm}"] = {"split9_chain": stat(y[:rows], ref), "fp32_mfma_layers": stat(z, ref)}
            out[f"backward_{name}_bias_gradient_of_{nm}"] = {"split9_chain": stat(b, ref.sum(0)), "fp32_mfma_layers": stat(zb, ref.sum(0))}
    # weight gradients: the six layers at the update's full batch (the error of a sum over the batch grows with it), ELU outputs x N(0, 0.01) gradients
    from booster_gym_amd.utils.model import plan_wgrad_slices
    M = 98304
    six = [(128, 256, 256), (256, 256, 256), (256, 64, 61), (128, 128, 128), (128, 256, 256), (256, 64, 47)]
    g = torch.Generator(device="cpu").manual_seed(3)
    data = []
    for co, ci, cr in six:
        G = (torch.randn(M, co, generator=g) * 0.01).to("cuda")
        A = torch.zeros(M, ci); A[:, :cr] = torch.nn.functional.elu(torch.randn(M, cr, generator=g)); A = A.to("cuda")
        data.append((G, A))
    res = {}
    for kind, share in (("fp32_mfma", False), ("split9", True)):
        slices, tw = plan_wgrad_slices([(co, ci) for co, ci, _ in six], M, 256, share_rows=share)
        arr, keep = (_lib.WgradProblem * len(six))(), []
        for k, ((co, ci, cr), (G, A), sl) in enumerate(zip(six, data, slices)):
            dW, sc = torch.empty(co, cr, device="cuda"), torch.empty(sl * co * ci, device="cuda")
            keep.append((dW, sc))
            arr[k].G, arr[k].A, arr[k].dW, arr[k].scratch = G.data_ptr(), A.data_ptr(), dW.data_ptr(), sc.data_ptr()
            arr[k].M, arr[k].C_out, arr[k].C_in, arr[k].C_in_real, arr[k].slices, arr[k].tiles_per_workgroup = M, co, ci, cr, sl, tw[k]
        if share:
            _lib.check(lib.bg_mlp_weight_grad_group_split(arr, len(six), 9, st), "bg_mlp_weight_grad_group_split")
        else:
            _lib.check(lib.bg_mlp_weight_grad_group(arr, len(six), st), "bg_mlp_weight_grad_group")
        res[kind] = [dW for dW, _ in keep]
    for k, ((co, ci, cr), (G, A)) in enumerate(zip(six, data)):
        ref = G.double().t() @ A.double()[:, :cr]
        out[f"weight_gradient_{k}_{co}x{cr}"] = {"split9": stat(res["split9"][k], ref), "fp32_mfma": stat(res["fp32_mfma"][k], ref)}
    out["note"] = (f"{rows} rows of N(0, 1) inputs / N(0, 0.01) gradients through the two networks' layer shapes with 1 / sqrt(fan-in) weights (tests/test_gpu_mlp_chain_split*.py hold "
                   "the same comparison at 98,304 rows); weight gradients: 98,304 rows of ELU outputs x N(0, 0.01) gradients.  The bf16 MFMA's accumulator does not round to nearest "
                   "(it truncates: a sum comes out low): odd slabs of the chains and odd sub-ranges of the batch in the weight gradients accumulate the negated sums, which makes "
                   "the offset cancel in every sum over rows (DESIGN.md section 6.2)")
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

    # HIP events (on the launch stream) around the env-step launches, both networks' chained forward launches, both backward-data chains, the grouped
    # weight-gradient launch pair and (ranks of a group, also a world of one under BG_DIST_FORCE=1) every exchange of the collective path.  A timing
    # event is a marker the queue stops at: ~250 of them per iteration cost the loop 0.5 ms of its 24 (tools/loop_time.py: the same loop with none,
    # profiles/r05_bench_instrumentation_cost.txt).  They are therefore armed on every KERNEL_EVENTS_EVERY-th iteration of the timed region: the
    # roofline entries average the launches of those iterations, `value` and `phase_ms` cover all of them.
    wgrad_events, dp_events = [], {"moments": [], "bucket": [], "stats": []}

    def arm_kernel_events(on):
        runner.env.step_to = timed_step_to if on else orig_step_to
        runner._critic_tr.timed_layer = runner._actor_tr.timed_layer = (1, 2) if on else None
        runner._wgrad_group.timed_events = wgrad_events if on else None
        if runner.dp.active:
            runner.dp.timed_events = dp_events if on else None

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
        arm_kernel_events(k % KERNEL_EVENTS_EVERY == 0)
        runner.train_iteration(args.warmup + k)
    arm_kernel_events(False)
    barrier()
    wall = time.perf_counter() - t0
    runner._flush_log()  # the last iteration's scalars (outside the timed region: train() does the same after its loop)
    tw = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    wall = float(tw.item())

    log(f"timed region done: {wall:.3f}s for {args.steps} iterations")
    slow_phase = None
    if world > 1:  # the largest rollout / update time over the ranks (one line must show a straggler)
        ph = torch.tensor([sum(a.elapsed_time(b) for a, b, _ in phase_events) / len(phase_events), sum(b.elapsed_time(c) for _, b, c in phase_events) / len(phase_events)],
                          dtype=torch.float64, device=dev)
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        slow_phase = {"rollout": float(ph[0].item()), "update": float(ph[1].item())}
    if rank == 0:
        step_ms = sum(a.elapsed_time(b) for a, b in step_events) / max(len(step_events), 1)
        roll_ms = sum(a.elapsed_time(b) for a, b, _ in phase_events) / len(phase_events)
        upd_ms = sum(b.elapsed_time(c) for _, b, c in phase_events) / len(phase_events)
        it_ms = [a.elapsed_time(c) for a, _, c in phase_events]
        armed_ms = [v for k, v in enumerate(it_ms) if k % KERNEL_EVENTS_EVERY == 0]
        plain_ms = [v for k, v in enumerate(it_ms) if k % KERNEL_EVENTS_EVERY != 0]
        # exchanges of the collective path per mini-epoch, rank 0's view: collective + waiting for the slowest rank (advantage moments,
        # the 712 kB gradient bucket, loss / KL sums)
        ex = dp_events if runner.dp.active else {}
        ex_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / len(v) if v else 0.0) for k, v in ex.items()}
        ar_ms = ex_ms.get("bucket", 0.0)
        env_bytes = N * ENV_STEP_BYTES
        sim_gbs = env_bytes / (step_ms * 1e-3) / 1e9
        flops = gemm_flops_per_iteration(N, T, E)

        def chain_flop(tr, rows):
            """algorithmic flops of a network's chained forward launch: the REAL input columns (47 / 61 of the zero-padded 64)"""
            w = [l.weight.shape for l in tr.layers[:3]]
            return 2.0 * rows * sum(o * i for o, i in w)

        def solo_us(fn, reps=30):
            for _ in range(100):  # (the clock under a new full-chip kernel is a transient for the first tens of milliseconds: time the sustained rate)
                fn()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(reps):
                fn()
            g1.record(); torch.cuda.synchronize()
            return g0.elapsed_time(g1) / reps * 1e3

        lib = _lib.load()
        evc = [e for e in runner._critic_tr.timed_events if e[5] in ("chain", "chain_split")]
        eva = [e for e in runner._actor_tr.timed_events if e[5] in ("chain", "chain_split")]
        if not (evc and len(evc) == len(eva)):
            raise SystemExit("bench.py expects the chained forward kernels (the default path) on both networks")
        # The symbols with the largest total time of the iteration (profiles/<PMC_TAG>_bench_kernel_stats.csv) are the two networks' chained forward
        # launches.  They run side by side on two streams and share the machine by slabs, so neither launch's own duration prices its work:
        # the figure is the flops of both over the time from the first start to the last end of each mini-epoch's pair.
        spans, c_us, a_us = [], [], []
        for (c0, c1, *_), (a0, a1, *_) in zip(evc, eva):
            first = a0 if a0.elapsed_time(c0) >= 0 else c0
            last = c1 if a1.elapsed_time(c1) >= 0 else a1
            spans.append(first.elapsed_time(last)); c_us.append(c0.elapsed_time(c1)); a_us.append(a0.elapsed_time(a1))
        span_us, c_loop_us, a_loop_us = (sum(v) / len(v) * 1e3 for v in (spans, c_us, a_us))
        rows_c, rows_a = evc[0][2], eva[0][2]
        fl_c, fl_a = chain_flop(runner._critic_tr, rows_c), chain_flop(runner._actor_tr, rows_a)
        full = rows_c == (T + 1) * 4096
        split_fwd = runner._critic_tr._chain_split() and runner._actor_tr._chain_split()
        split_bwd = runner._critic_tr._chain_split_bwd() and runner._actor_tr._chain_split_bwd()
        kc_name, ka_name = ("mlp_chain_split_fwd_kernel<2>", "mlp_chain_split_fwd_kernel<1>") if split_fwd else ("mlp_chain_fwd_kernel<2>", "mlp_chain_fwd_kernel<1>")
        one_launch = evc[0][0] is eva[0][0]  # both networks in ONE grid (Runner._one_stream): the trainers noted the same event pair
        tr_c = (pmc_traffic(kc_name), pmc_traffic(ka_name)) if full else (None, None)
        tr_group = pmc_traffic("mlp_chain_split_fwd_kernel<0>") if full and one_launch else None  # (summarised at its largest grid: the update's launch)
        tf = (fl_c + fl_a) / (span_us * 1e-6) / 1e12
        fwd_peak = SPLIT9_PEAK_TF if split_fwd else MFMA_F32_PEAK_TF
        arith = ("fp32 operands as exact three-way bf16 splits, all 9 cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, bg_mlp_chain_split.hip"
                 if split_fwd else "fp32 MFMA 32x32x2, bg_mlp_chain.hip")
        headline = {"kernel": (f"mlp_chain_split_fwd_kernel<0>: the critic's and the actor's three fused Linear+bias+ELU hidden layers in ONE launch "
                               f"([{rows_c}x61] -> 256 -> 256 -> 128 on {int(runner._critic_tr.chain_workgroups)} workgroups and [{rows_a}x47] -> 256 -> 128 -> 128 on "
                               f"{int(runner._actor_tr.chain_workgroups)}, one per CU, persistent over their slabs; activations handed on in registers, {arith}, hand-written HIP)")
                              if one_launch else
                              (f"{kc_name} + {ka_name}: the critic's and the actor's three fused Linear+bias+ELU hidden layers, one launch per "
                               f"network ([{rows_c}x61] -> 256 -> 256 -> 128 and [{rows_a}x47] -> 256 -> 128 -> 128, activations handed on in registers, {arith}, "
                               "hand-written HIP); the two launches of a mini-epoch overlap on two streams"),
                    "bound": "mfma", "achieved": tf, "peak": fwd_peak, "unit": "TFLOP/s", "frac": tf / fwd_peak,
                    "peak_note": (f"peak = {MFMA_BF16_PEAK_TF} TF/s dense bf16 MFMA / 9 products per fp32 x fp32 product = {SPLIT9_PEAK_TF:.1f} TF/s of fp32-exact flops; "
                                  "`achieved` counts the algorithmic fp32 flops once") if split_fwd else "fp32-input MFMA dense peak",
                    "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TF, "fp32_mfma_peak": MFMA_F32_PEAK_TF,
                    "traffic": tr_group if one_launch else ((tr_c[0] + tr_c[1]) if all(tr_c) else None), "traffic_source": PMC_SOURCE,
                    "avg_launch_us": span_us, "algorithmic_flops_per_launch": fl_c + fl_a,
                    "note": ("flops of both networks (real input columns) / duration of the launch, HIP events on the launch stream inside the timed loop"
                             if one_launch else
                             "flops of both launches (real input columns) / time from the first start to the last end of the pair, HIP events on the two launch "
                             "streams inside the timed loop; the pair's own durations are in per_kernel_in_the_loop"),
                    "per_kernel_in_the_loop": ({"mlp_chain_split_fwd_kernel<0>": {"avg_launch_us": span_us, "algorithmic_flops": fl_c + fl_a, "traffic": tr_group}}
                                               if one_launch else
                                               {kc_name: {"avg_launch_us": c_loop_us, "algorithmic_flops": fl_c, "traffic": tr_c[0]},
                                                ka_name: {"avg_launch_us": a_loop_us, "algorithmic_flops": fl_a, "traffic": tr_c[1]}})}
        layer_fwd = None if one_launch else {"kernel": f"{kc_name}: the critic's chained forward launch, [{rows_c}x61] -> 256 -> 256 -> 128", "bound": "mfma",
                     "achieved": fl_c / (c_loop_us * 1e-6) / 1e12, "peak": fwd_peak, "unit": "TFLOP/s", "frac": fl_c / (c_loop_us * 1e-6) / 1e12 / fwd_peak,
                     "frac_of_fp32_mfma_peak": fl_c / (c_loop_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF,
                     "traffic": tr_c[0], "traffic_source": PMC_SOURCE, "avg_launch_us": c_loop_us, "algorithmic_flops_per_launch": fl_c,
                     "note": "timed inside the loop, where the actor's launch runs beside it on the second stream"}
        if not args.no_extra:
            # the same launches with nothing beside them (hidden layers only: the critic's value head is left out of the stand-alone launch)
            from booster_gym_amd.utils.model import MLPTrainer as _MLPT
            alone = {}
            for name, tr, fl in ((kc_name, runner._critic_tr, fl_c), (ka_name, runner._actor_tr, fl_a)):
                keep, tr.value_head = tr.value_head, None
                d = tr._chain_descriptor()
                tr.value_head = keep
                us = solo_us(lambda: _MLPT.launch_chain([d]))
                alone[name] = {"avg_launch_us": us, "achieved": fl / (us * 1e-6) / 1e12, "frac": fl / (us * 1e-6) / 1e12 / fwd_peak,
                               # the loop's own launch geometry: a persistent grid on this network's share of the CUs (Runner._plan_chain_split), so
                               # "alone" = nothing beside it on the chip, not "on all CUs"
                               "workgroups": int(tr.chain_workgroups) or "one per 128-row slab"}
            headline["alone_on_the_gpu"] = alone
            if layer_fwd is not None:
                layer_fwd["alone_on_the_gpu"] = alone[kc_name]
        wg_ev = wgrad_events
        wgrad = None
        if wg_ev:  # all six hidden-layer weight gradients of both networks: one launch pair per mini-epoch, alone on the GPU
            wus = sum(a.elapsed_time(b) for a, b, *_ in wg_ev) / len(wg_ev) * 1e3
            wfl = wg_ev[0][2]
            split_wg = bool(getattr(runner._wgrad_group, "split", 0))
            wg_peak = SPLIT9_PEAK_TF if split_wg else MFMA_F32_PEAK_TF
            wg_kernel = "mlp_wgrad_group_split_kernel<9>" if split_wg else "mlp_wgrad_group_kernel"
            wgrad = {"kernel": f"{wg_kernel}: dW = G^T A of all six hidden layers of both networks in one launch (its fixed-order finish runs inside the tail's first "
                               "launch), " + " + ".join(f"[{co}x{m}]x[{m}x{ci}]" for m, co, ci in wg_ev[0][3]) +
                               (", fp32 operands as exact three-way bf16 splits, all 9 products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, the sub-ranges of the batch "
                                "alternating the sign of the accumulation (hand-written HIP, bg_wgrad_split.hip)" if split_wg else ", fp32 MFMA 32x32x2 (hand-written HIP, bg_wgrad.hip)"),
                     "bound": "mfma", "achieved": wfl / (wus * 1e-6) / 1e12, "peak": wg_peak, "unit": "TFLOP/s",
                     "frac": wfl / (wus * 1e-6) / 1e12 / wg_peak, "frac_of_fp32_mfma_peak": wfl / (wus * 1e-6) / 1e12 / MFMA_F32_PEAK_TF,
                     "traffic": pmc_traffic(wg_kernel) if N == 4096 else None,
                     "traffic_source": PMC_SOURCE, "avg_launch_us": wus, "algorithmic_flops_per_launch": wfl,
                     "note": "timed inside the loop with HIP events on its stream; it runs after both backward chains, alone on the GPU"}
        # HBM traffic per launch comes from PMC counters, which rocprofv3 collects in separate passes of the same command (tools/profile.sh
        # -> profiles/<PMC_TAG>_*_pmc.json, FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes); the JSON line names the file it cites
        traffic = pmc_traffic("env_step_kernel", which="env") if N == 4096 else None
        env_valu_busy = pmc_ratio("env_step_kernel", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", which="env") if N == 4096 else None
        # backward-data chains (dX = G W of the hidden layers, ELU' and bias-gradient sums in the epilogues): the critic's on the side stream, the actor's
        # on the main stream; priced like the forward pair, over the span from the first start to the last end of each mini-epoch's two chains
        bwc = [e for e in runner._critic_tr.timed_events if e[5] == "backward"]
        bwa = [e for e in runner._actor_tr.timed_events if e[5] == "backward"]
        backward = None
        if bwc and len(bwc) == len(bwa):
            bspans = []
            for (c0, c1, *_), (a0, a1, *_) in zip(bwc, bwa):
                first = a0 if a0.elapsed_time(c0) >= 0 else c0
                last = c1 if a1.elapsed_time(c1) >= 0 else a1
                bspans.append(first.elapsed_time(last))
            bus = sum(bspans) / len(bspans) * 1e3
            bfl = bwc[0][3] + bwa[0][3]
            bwd_peak = SPLIT9_PEAK_TF if split_bwd else MFMA_F32_PEAK_TF
            backward = {"kernel": ("mlp_chain_split_bwd_kernel<0> / <1>: the backward-data pass of each network's hidden layers as ONE launch (G2 = (G3 W3) elu'(A2), "
                                   "G1 = (G2 W2) elu'(A1), bias-gradient column sums; fp32 operands as exact three-way bf16 splits, 9 products, fp32 accumulation, "
                                   "bg_mlp_chain_split_bwd.hip), " + ("both networks in ONE launch that shares the chip by CUs" if bwc[0][0] is bwa[0][0] else "two launches on two streams")) if split_bwd else
                                  ("mlp_fwd_kernel<256,2,2> / <128,2,2> / <128,2,1>: the backward-data GEMMs of both networks' hidden layers (dX = G W with ELU' and the "
                                   "bias-gradient column sums in the epilogue), two chains of two launches on two streams"),
                        "bound": "mfma", "achieved": bfl / (bus * 1e-6) / 1e12, "peak": bwd_peak, "unit": "TFLOP/s", "frac": bfl / (bus * 1e-6) / 1e12 / bwd_peak,
                        "frac_of_fp32_mfma_peak": bfl / (bus * 1e-6) / 1e12 / MFMA_F32_PEAK_TF,
                        "avg_launch_us": bus, "algorithmic_flops_per_launch": bfl,
                        "per_chain_in_the_loop_us": {"critic": sum(a.elapsed_time(b) for a, b, *_ in bwc) / len(bwc) * 1e3,
                                                     "actor": sum(a.elapsed_time(b) for a, b, *_ in bwa) / len(bwa) * 1e3},
                        "note": "flops of both chains / time from the first start to the last end of the pair (one launch: its duration), HIP events on the launch "
                                "stream(s) inside the timed loop.  The kernel runs against the CU's vector-memory pipeline, not the matrix pipe "
                                "(profiles/r06_chain_split_bwd_stamps_and_ablations.txt)"}
        out = {
            "metric": "env-steps/sec (whole node), PPO rollout+update, T1 4096 envs/GPU",
            "value": world * N * T * args.steps / wall, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (random-init policy, seeded domain randomisation)",
            "config": {"workload": f"T1 {args.terrain} terrain, {N} envs/GPU, horizon {T}, {E} mini-epochs, full batch (BASELINE.json configs[1])",
                       "envs_per_gpu": N, "parallelism": f"dp{world}",
                       "gemm_arithmetic": ({9: "fp32 operands as exact 3-way bf16 splits, 9 products, fp32 accumulate (BG_GEMM_SPLIT=9, per-layer kernels)",
                                            6: "fp32 operands as exact 3-way bf16 splits, 6 largest products, fp32 accumulate (BG_GEMM_SPLIT=6, per-layer kernels)"}[split_mode]
                                           if split_mode else
                                           ("hidden-layer forward" + (", backward-data" if split_bwd else "") +
                                            (" and weight-gradient" if getattr(runner._wgrad_group, "split", 0) else "") + " GEMMs: every fp32 operand the EXACT sum of three bf16 numbers "
                                            "(8 + 8 + 8 significant bits), all 9 cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation -- products exact as in an fp32 FMA "
                                            "chain, measured error against float64 at or below the fp32-MFMA kernels' (gemm_errors_vs_float64)"
                                            + ("" if split_bwd else "; backward-data: fp32 MFMA") + ("" if getattr(runner._wgrad_group, "split", 0) else "; weight gradients: fp32 MFMA")
                                            + "; heads, rollout actor: fp32 VALU / fp32 MFMA (v_mfma_f32_16x16x4_f32).  `fp32_mfma_loop`: the same loop with everything on the fp32 "
                                            "matrix pipe") if split_fwd else
                                           "fp32 MFMA (v_mfma_f32_32x32x2_f32)")},
            "ppo_iters_per_s": args.steps / wall,
            "phase_ms": {"rollout": roll_ms, "update": upd_ms, "all_reduce_ms": ar_ms},
            "kernel_events": {"armed_on_every": KERNEL_EVENTS_EVERY, "iterations_armed": len(armed_ms), "of": args.steps,
                              "rollout_plus_update_ms": {"armed": sum(armed_ms) / max(len(armed_ms), 1), "unarmed": sum(plain_ms) / len(plain_ms) if plain_ms else None},
                              "note": "HIP timing events around the env-step, chained-forward, backward-data and weight-gradient launches (what the roofline "
                                      "entries average) are markers the queues stop at; they are recorded on a subset of the timed iterations, `value`, "
                                      "`ms_per_step` and `phase_ms` cover all of them; tools/loop_time.py runs the loop with none"},
            "exchange_ms_per_mini_epoch": ex_ms or None,
            "roofline": headline,
            "roofline_layer_forward": layer_fwd,
            "roofline_env_step": {"kernel": "env_step_kernel (hand-written HIP: 10 ABA substeps + task logic, one launch per env-step)", "bound": "hbm",
                                  "achieved": sim_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sim_gbs / HBM_PEAK_GBS, "traffic": traffic,
                                  "traffic_source": PMC_SOURCE, "avg_launch_us": step_ms * 1e3, "algorithmic_bytes_per_launch": env_bytes,
                                  "valu_busy_share_of_wave_cycles": env_valu_busy,
                                  "note": f"issue-latency-bound at 128 waves, one per SIMD on half of the CUs: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES in profiles/{PMC_TAG}_env_pmc.json "
                                          "(valu_busy_share_of_wave_cycles)"},
            "roofline_update": {"bound": "mfma", "achieved": flops / (upd_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": flops / (upd_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                                "note": "all actor+critic GEMM flops of the update phase / update-phase wall time (which also holds GAE, loss, ELU, Adam)"},
            "nonfinite_resets": runner.nonfinite_resets_total,
        }
        if wgrad is not None:
            out["roofline_wgrad"] = wgrad
        if backward is not None:
            out["roofline_backward"] = backward
        if world > 1:
            # what one SCALE record needs to tell a bad queue mapping or a fallback communicator from a slow fabric
            out["multi_rank"] = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "own_rccl": runner.dp.comm is not None,
                                 "backend": runner.dp.backend, "exchange_ms_per_mini_epoch": ex_ms or None, "phase_ms_slowest_rank": slow_phase,
                                 "note": "exchange times are rank 0's (collective + waiting for the slowest rank); phase_ms_slowest_rank: the largest rollout / update time over the ranks"}
        if world == 1 and not args.no_extra:
            try:
                # The same loop, same run, same timing with BOTH chains on the fp32 matrix pipe (last round's headline path: bg_mlp_chain.hip forward,
                # bg_mlp_layer_backward per layer): what a reader who does not accept the 9-product arithmetic as fp32 falls back on.  NOT part of `value`.
                from booster_gym_amd.utils.model import MLPTrainer

                runner.rollout, runner.update, runner.env.step_to = orig_rollout, orig_update, orig_step_to
                runner._critic_tr.timed_layer, runner._actor_tr.timed_layer, runner._wgrad_group.timed_events = None, None, None
                it0 = args.warmup + args.steps

                def timed_loop(it0):
                    runner.invalidate()
                    for _ in range(2):
                        runner.train_iteration(it0); it0 += 1
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for _ in range(args.steps):
                        runner.train_iteration(it0); it0 += 1
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - ts
                    return it0, {"value": N * T * args.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / args.steps * 1e3}

                keep = (MLPTrainer.CHAIN_SPLIT, MLPTrainer.CHAIN_SPLIT_BWD, MLPTrainer.WGRAD_SPLIT)
                MLPTrainer.CHAIN_SPLIT = MLPTrainer.CHAIN_SPLIT_BWD = False
                it0, fp32 = timed_loop(it0)
                MLPTrainer.CHAIN_SPLIT, MLPTrainer.CHAIN_SPLIT_BWD = True, False
                it0, fwd_only = timed_loop(it0)
                MLPTrainer.CHAIN_SPLIT, MLPTrainer.CHAIN_SPLIT_BWD, MLPTrainer.WGRAD_SPLIT = keep[0], keep[1], 0
                it0, fp32_wg = timed_loop(it0)
                out["fp32_mfma_weight_gradients_loop"] = dict(fp32_wg, gemm_arithmetic="split forward and backward chains, fp32-MFMA weight gradients: BG_WGRAD_SPLIT=0 "
                                                                                       "(the headline path until late round 6)")
                MLPTrainer.CHAIN_SPLIT, MLPTrainer.CHAIN_SPLIT_BWD, MLPTrainer.WGRAD_SPLIT = keep
                one_stream, runner._one_stream = runner._one_stream, False
                it0, two_streams = timed_loop(it0)
                runner._one_stream = one_stream
                it0, again = timed_loop(it0)
                out["two_launches_on_two_streams_loop"] = dict(two_streams, note="the headline arithmetic with the critic's and the actor's chains as separate launches on "
                                                                                 "two streams (BG_ONE_STREAM=0: the form until mid round 6)")
                runner._flush_log()
                fp32["gemm_arithmetic"] = "fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere: BG_CHAIN_SPLIT=0 (with it the weight gradients take the fp32 launch too)"
                out["fp32_mfma_loop"] = fp32
                out["split_forward_only_loop"] = dict(fwd_only, gemm_arithmetic="split forward chain, fp32-MFMA backward layers: BG_CHAIN_SPLIT_BWD=0")
                out["headline_loop_again"] = dict(again, note="the headline configuration once more behind the two above, uninstrumented (no timing events): what `value` is to be compared with")
            except Exception as ex:
                out["fp32_mfma_loop"] = {"error": repr(ex)}
            try:
                out["gemm_errors_vs_float64"] = gemm_error_pairs()
            except Exception as ex:
                out["gemm_errors_vs_float64"] = {"error": repr(ex)}
            try:
                # Opt-in form of the layer kernels, measured beside the headline and NOT part of `value`: BG_GEMM_SPLIT (bg_mlp_split.hip) runs the
                # fp32 x fp32 products of the hidden-layer forward / backward GEMMs on the bf16 matrix pipe, every fp32 operand split EXACTLY into
                # three bf16 numbers (all 9 cross products: no rounding of the products, fp32 accumulation; 6: the three smallest dropped);
                # so does the grouped weight-gradient launch (bg_wgrad_split.hip).  Same loop, same workload, same timing as `value`.
                split = {}
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
                split["note"] = ("opt-in (BG_GEMM_SPLIT=9|6), not the headline and superseded by the chained kernels: one launch per layer (forward / backward / "
                                 "grouped weight gradients) as exact hi/mid/lo bf16 splits on v_mfma_f32_32x32x16_bf16, fp32 accumulate; products_6 drops the three "
                                 "smallest cross products and is NOT fp32-exact; tests/test_gpu_mlp_split.py holds the errors against float64")
                out["opt_in_split_bf16_layers"] = split
            except Exception as ex:
                out["opt_in_split_bf16_layers"] = {"error": repr(ex)}
            try:
                # The other single-GPU configurations of BASELINE.json, the same loop (Runner.train_iteration) timed the same way after the headline;
                # not part of `value`.  configs[2]: the shipped rough terrain with the command curriculum on; configs[4]: full randomisation,
                # 16,384 envs, simulator state stored in fp16.
                others = {}
                for key, over, what in (
                        ("configs[2]", {"env.num_envs": 4096, "terrain.type": "trimesh", "commands.curriculum": True},
                         "T1 heightfield terrain (utils/terrain.py) with the command curriculum, 4096 envs"),
                        ("configs[4]", {"env.num_envs": 16384, "terrain.type": "trimesh", "sim.state_dtype": "fp16"},
                         "T1 full domain randomisation (mass / friction / latency / push), 16384 envs per GPU, fp16 state")):
                    c2 = load_cfg("T1", dict({"basic.seed": 42}, **over))
                    c2["runner"]["save_interval"] = 10 ** 9
                    r2 = Runner(cfg=c2)
                    r2.begin_training(Recorder(c2, root=log_root, rank=rank))
                    for w in range(args.warmup):
                        r2.train_iteration(w)
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for k in range(args.steps):
                        r2.train_iteration(args.warmup + k)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - ts
                    r2._flush_log()
                    others[key] = {"workload": what, "value": r2.env.num_envs * T * args.steps / dt, "unit": "env-steps/s", "ms_per_step": dt / args.steps * 1e3,
                                   "ppo_iters_per_s": args.steps / dt, "steps": args.steps, "warmup": args.warmup, "nonfinite_resets": r2.nonfinite_resets_total}
                    del r2
                out["other_configs"] = others
            except Exception as ex:
                out["other_configs"] = {"error": repr(ex)}
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
