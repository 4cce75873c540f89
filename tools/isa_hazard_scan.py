"""Static check of the MFMAs that bg_mlp_chain_split_bwd.hip issues through inline asm (the compiler does not know that those statements are MFMAs and
therefore does not keep the wait states a vector write needs before an MFMA reads the register): compiles the file to gfx950 assembly and, for every
inline-asm v_mfma, counts the wait states between it and the nearest earlier vector instruction that writes one of its operand registers (an instruction
in between = one wait state, s_nop N = N + 1).  Fewer than two = a hazard.  No GPU needed.
    python tools/isa_hazard_scan.py [extra hipcc flags]   -> one line per kernel; exit code 1 if any hazard"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "booster_gym_amd", "csrc", "bg_mlp_chain_split_bwd.hip")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -fassociative-math -freciprocal-math -fno-trapping-math -mllvm -amdgpu-sched-strategy=max-ilp".split()
NEED = 2


def regs(tok):
    tok = tok.strip()
    m = re.match(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([av])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def scan(extra=()):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-S", "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.unlink(out)
    report = {}
    for f in re.split(r"\n(?=_Z[\w]+:\s)", txt):
        name = f.split(":", 1)[0]
        if not name.startswith("_Z") or "mlp_chain_split_bwd_kernel" not in name:
            continue
        ins, in_asm = [], False
        for l in f.splitlines():
            l = l.strip()
            if l.startswith(";;#ASMSTART"):
                in_asm = True
            elif l.startswith(";;#ASMEND"):
                in_asm = False
            elif l and not l.startswith((";", ".", "//")) and not l.endswith(":"):
                ins.append((l, in_asm))
        n_mfma, hazards, worst = 0, [], None
        for k, (l, a) in enumerate(ins):
            if not (a and l.startswith("v_mfma")):
                continue
            n_mfma += 1
            reads = set().union(*[regs(t) for t in l.split(None, 1)[1].split(",")[1:]])
            states = 0
            for back in range(1, 8):
                p = ins[k - back][0]
                if p.startswith("s_nop"):
                    states += int(p.split()[1]) + 1
                    continue
                if p.startswith("v_") and not p.startswith("v_mfma") and regs(p.split(None, 1)[1].split(",")[0]) & reads:
                    worst = states if worst is None else min(worst, states)
                    if states < NEED:
                        hazards.append((p, l, states))
                    break
                states += 1
                if states >= NEED + 2:
                    break
        # the asm loads of the next slab's input rows: nothing may touch their destination registers before the barrier of the next chunk's top (the
        # counted wait that makes them valid sits in front of it)
        early = []
        for k, (l, a) in enumerate(ins):
            if not (a and l.startswith("global_load_dwordx4 v")):
                continue
            dst = regs(l.split(None, 1)[1].split(",")[0])
            for l2, _ in ins[k + 1:]:
                if l2.startswith("s_barrier"):
                    break
                toks = re.findall(r"[av]\[\d+:\d+\]|[av]\d+", l2)
                if any(regs(t) & dst for t in toks):
                    early.append((l, l2))
                    break
        report[name] = (n_mfma, hazards, worst, early)
    return report


if __name__ == "__main__":
    rep = scan(sys.argv[1:])
    bad = 0
    for name, (n, hz, worst, early) in rep.items():
        print(f"{name}: {n} inline-asm MFMAs, {len(hz)} with fewer than {NEED} wait states behind a vector write of an operand (closest: {worst}); "
              f"{len(early)} asm loads whose registers are touched before the next chunk's barrier")
        for p, l, s in hz[:6]:
            print(f"    {s} wait states: {p}  ->  {l}")
        for l, l2 in early[:6]:
            print(f"    {l}  touched by  {l2}")
        bad += len(hz) + len(early)
    sys.exit(1 if bad else 0)
