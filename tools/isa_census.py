"""Itemised instruction budget of the ABA kernel (forward_dynamics_kernel<false>): compiles bg_sim.hip to gfx950 assembly with -DBG_ISA_PHASES
(scheduling barriers + named markers between the phases of bg_dyn.h) and counts the VALU instructions between markers, in emission order.
No GPU needed.  The marked build schedules differently from the product build (the markers pin phase boundaries), so its total differs by a few
per cent from the product kernel's; the product kernel's own total is printed next to it.
    python tools/isa_census.py [kernel-name-substring] -> table on stdout, JSON in gpurun_out/isa_census.json"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "booster_gym_amd", "csrc", "bg_sim.hip")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -fassociative-math -freciprocal-math -fno-trapping-math -mllvm -amdgpu-sched-strategy=max-ilp".split()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[0] if args else "forward_dynamics_kernelILb0"
PLANE = ["-DBG_CENSUS_PLANE"] if "--plane" in sys.argv else []  # flat ground: leave the (never executed) height-field lookups out of the count
if "--t1" in sys.argv:   # the T1's links 1, 2, 4 sit on their parent's z axis: resolve the wave-uniform zmask branches as a T1 launch takes them
    PLANE = PLANE + ["-DBG_CENSUS_ZMASK=22"]
if "--generic" in sys.argv:  # ... or as a model with no such link takes them
    PLANE = PLANE + ["-DBG_CENSUS_ZMASK=0"]


def build(extra):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.unlink(out)
    for f in re.split(r"\n(?=_Z[\w]+:\s)", txt):
        if want in f.split(":", 1)[0]:
            return f
    raise SystemExit(f"kernel {want} not found")


def classify(op):
    if op.startswith(("v_fma", "v_fmac", "v_mac", "v_mad")): return "fma"
    if op.startswith("v_mul_f32"): return "mul"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")): return "add"
    if op.startswith(("v_sin", "v_cos", "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log")): return "transcendental"
    if op.startswith("v_pk_"): return "packed"
    return "other_valu"


def census(body):
    phases, cur = collections.OrderedDict(), "prologue"
    seen = collections.Counter()
    for line in body.split("\n"):
        l = line.strip()
        m = re.match(r";\s*BG_PHASE\s+(\w+)", l)
        if m:
            seen[m.group(1)] += 1
            cur = m.group(1) + (f"[{seen[m.group(1)] - 1}]" if m.group(1).endswith("_link") else "")
            continue
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        d = phases.setdefault(cur, collections.Counter())
        if op.startswith("v_"):
            d["valu"] += 1; d[classify(op)] += 1
        elif op.startswith("ds_"): d["lds"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")): d["vmem"] += 1
        elif op.startswith("s_"): d["salu"] += 1
    return phases


marked = census(build(["-DBG_ISA_PHASES"] + PLANE))
plain = census(build(PLANE))
tot = sum(d["valu"] for d in marked.values())
print(f"{'phase':34s} {'VALU':>6s} {'fma':>6s} {'mul':>6s} {'add':>6s} {'transc':>6s} {'other':>6s} {'lds':>5s} {'vmem':>5s}")
for k, d in marked.items():
    print(f"{k:34s} {d['valu']:6d} {d['fma']:6d} {d['mul']:6d} {d['add']:6d} {d['transcendental']:6d} {d['other_valu'] + d['packed']:6d} {d['lds']:5d} {d['vmem']:5d}")
print(f"{'TOTAL (marked build)':34s} {tot:6d}")
print(f"{'TOTAL (product build, no markers)':34s} {sum(d['valu'] for d in plain.values()):6d}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"kernel": want, "marked": {k: dict(v) for k, v in marked.items()}, "total_marked_valu": tot, "total_product_valu": sum(d["valu"] for d in plain.values())},
          open(os.path.join(ROOT, "gpurun_out", "isa_census.json"), "w"), indent=1)
