#!/bin/bash
# MFMA-pipe utilisation of the whole bench loop: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over every kernel of `bench.py --steps 5 --warmup 2`
# (run on the GPU box: gpurun -- bash tools/mfma_util.sh <tag>); writes gpurun_out/<tag>_bench_mfma_pmc.json
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_mfma.log 2>&1
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = sorted(glob.glob(f"{R}/gpurun_out/prof_{tag}_mfma/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
busy, gui, calls = collections.Counter(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[:60]
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[name] += float(r["Counter_Value"]); calls[name] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[name] += float(r["Counter_Value"])
line = json.loads([l for l in open(f"{R}/gpurun_out/prof_{tag}_mfma.log") if l.startswith("{")][-1])
iters = 7
total_busy = sum(busy.values()) / iters
upd_ms = float(os.environ.get("BG_UNPROFILED_UPDATE_MS", line["phase_ms"]["update"]))  # the counter pass serialises kernels: pass the update time of an unprofiled run
# effective clock from the counters of the MFMA kernels themselves: GRBM_GUI_ACTIVE is summed over the 8 XCDs
out = {"command": "tools/mfma_util.sh (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra)",
       "iterations_profiled": iters, "update_ms_per_iteration_in_this_run": upd_ms,
       "mfma_busy_cycles_per_iteration_sum_over_1024_simds": total_busy,
       "mfma_pipe_utilisation_of_the_update_phase_at_2.4GHz": total_busy / 1024 / (upd_ms * 1e-3 * 2.4e9),
       "per_kernel_busy_cycles_per_iteration": {k: v / iters for k, v in busy.most_common(12)},
       "per_kernel_calls_per_iteration": {k: calls[k] / iters for k, _ in busy.most_common(12)}}
json.dump(out, open(f"{R}/gpurun_out/{tag}_bench_mfma_pmc.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in list(out)[1:5]}))
PY
