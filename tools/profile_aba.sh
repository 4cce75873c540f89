#!/bin/bash
# rocprofv3 passes of the ABA launch alone (tools/aba_only.py: forward dynamics of 1 M envs, standing pose + 0.1 rad joint noise):
# kernel trace + stats, then HBM traffic and SQ counters in separate --pmc passes with --kernel-trace only (as the pool requires).
#   gpurun -- bash tools/profile_aba.sh r03   -> gpurun_out/<tag>_aba_kernel_stats.csv, gpurun_out/<tag>_aba_pmc.json
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_${TAG}_aba
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/aba_only.py 1048576 0.1"
# the trace run launches 1200 times: the kernel's sustained rate (the clock is a transient for the first ~160 launches, tools/aba_series.py)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD 1200 > $OUT.trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT.pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT.pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT.pmc3.log 2>&1
# ... and the packed form of the same launch (bg_env_forward_dynamics_packed: forward_dynamics_pk_kernel) into the same summary
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pk -- $CMD 1200 1 > $OUT.trace_pk.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD 12 1 > $OUT.pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD 12 1 > $OUT.pmc5.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD 12 1 > $OUT.pmc6.log 2>&1
cp $(ls -t $OUT/trace_pk/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_aba_packed_kernel_stats.csv
cp $(ls -t $OUT/trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_aba_kernel_stats.csv
python3 - "$OUT" "$R/gpurun_out/${TAG}_aba_pmc.json" <<'PY'
import collections, csv, glob, json, os, sys
root, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"pass {p} left no counter CSV")
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if "forward_dynamics" in name or "aba_compact" in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "tools/profile_aba.sh: rocprofv3 --kernel-trace --pmc <one counter set per run> -- python3 tools/aba_only.py 1048576 0.1 (12 launches)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch as rocprofv3 reports them; dword-per-lane accesses (uncalibrated width on gfx950, MI355X_MICROARCH.md "
                "HBM section: only 16-B-per-lane streams are calibrated), so hbm_bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 is indicative.  SQ_* summed over all waves.",
       "kernels": {}}
for k, cs in sorted(acc.items()):
    e = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes"] = (e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
    if "SQ_INSTS_VALU" in e and "SQ_WAVES" in e:
        e["valu_per_wave"] = e["SQ_INSTS_VALU"]["mean"] / e["SQ_WAVES"]["mean"]
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {c: (round(v["mean"]) if isinstance(v, dict) else round(v, 1)) for c, v in cs.items()} for k, cs in out["kernels"].items()}, indent=1))
PY
