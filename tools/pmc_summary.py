"""Summarise the rocprofv3 --pmc passes of tools/profile.sh: mean counter value per kernel and launch.
  gpurun_out/prof_<tag>/pmc_{fetch,write,sq}    (tools/prof_env.py 4096 plane)   -> gpurun_out/<tag>_env_pmc.json
  gpurun_out/prof_<tag>/pmcb_{fetch,write}      (bench.py --steps 5 --warmup 2)  -> gpurun_out/<tag>_bench_pmc.json
Fails when an expected pass left no counter CSV (copy the two JSON files into profiles/ to keep them)."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(R, "gpurun_out", "prof_" + tag)
UNITS = ("FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 (KiB per launch).  MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of a "
         "wide (16 B per lane) coalesced streaming read -> `hbm_bytes` = 2 x FETCH_SIZE + WRITE_SIZE for the kernels flagged wide16 (the MFMA layer "
         "kernels, whose loads and stores are all 16 B per lane), FETCH_SIZE + WRITE_SIZE for the dword-per-lane env kernels (uncalibrated width).  "
         "SQ_* summed over all waves, WAVE_CYCLES / ACTIVE / WAIT in quad-cycles")


def summarise(passes, keep, command):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in passes:
        files = glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise SystemExit(f"pmc_summary: pass {p} of tools/profile.sh {tag} left no counter CSV under {root}/{p}")
        for f in files:
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                if any(k in name for k in keep):
                    acc[(name, int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # a kernel launched with several grid sizes (the chained forward: whole-batch launches in the update, 64-slab launches during the rollout) is listed
    # per grid size; the plain name stands for its LARGEST grid (what the bench's roofline entries cite)
    grids = collections.defaultdict(list)
    for name, g in acc:
        grids[name].append(g)
    named = {}
    for (name, g), cs in acc.items():
        if g == max(grids[name]):
            named[name] = cs
        if len(grids[name]) > 1:
            named[f"{name} [grid={g}]"] = cs
    ks = {}
    for k, cs in sorted(named.items()):
        e = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            wide = any(s in k for s in ("mlp_fwd_kernel", "mlp_chain", "mlp_wgrad"))
            e["wide16"] = wide
            e["hbm_bytes"] = ((2.0 if wide else 1.0) * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
        ks[k] = e
    return {"command": command, "units": UNITS, "kernels": ks}


env = summarise(["pmc_fetch", "pmc_write", "pmc_sq"], ("env_step", "forward_dynamics", "actor_sample"),
                f"tools/profile.sh {tag}: rocprofv3 --kernel-trace --pmc <one counter set per run> -- python3 tools/prof_env.py 4096 plane")
bench = summarise(["pmcb_fetch", "pmcb_write"], ("mlp_", "head", "gae", "adam", "env_step", "actor_sample", "Cijk", "reduce_kernel"),
                  f"tools/profile.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra")
json.dump(env, open(os.path.join(R, "gpurun_out", tag + "_env_pmc.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(R, "gpurun_out", tag + "_bench_pmc.json"), "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"]) if isinstance(v, dict) else v for c, v in cs.items()} for k, cs in {**env["kernels"], **bench["kernels"]}.items()}, indent=1))
