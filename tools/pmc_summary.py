"""Summarise the rocprofv3 --pmc passes of tools/profile.sh: mean counter value per kernel and launch -> profiles/<tag>_env_pmc.json."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "prof_" + tag)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "env_step" in name or "forward_dynamics" in name or "mlp_fwd" in name or "head" in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "tools/profile.sh %s (rocprofv3 --kernel-trace --pmc <counter set>, one set per run; tools/prof_env.py 4096 plane)" % tag,
       "units": "FETCH_SIZE/WRITE_SIZE as reported by rocprofv3 (KiB per launch; dword-per-lane accesses, uncalibrated: see MI355X_MICROARCH.md HBM section); "
                "SQ_* summed over all waves, WAVE_CYCLES/ACTIVE/WAIT in quad-cycles",
       "kernels": {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}}
dst = os.path.join(os.path.dirname(root), "..", "profiles", tag + "_env_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"]) for c, v in cs.items()} for k, cs in out["kernels"].items()}, indent=1))
