#!/bin/bash
# Where the SIMDs' issue time of the update phase goes, per kernel: rocprofv3 --pmc over bench.py --steps 3 --warmup 2 (kernels serialised by the counter
# pass: per-launch counts are those of the kernel alone).  gpurun -- bash tools/pipe_pmc.sh <tag>  ->  gpurun_out/<tag>_pipe_pmc.json
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv \
  -d $R/gpurun_out/prof_${TAG}_pipe -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_pipe.log 2>&1
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = sorted(glob.glob(f"{R}/gpurun_out/prof_{tag}_pipe/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()[:70]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    n = len(cs["SQ_INSTS_VALU"])
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get("SQ_INSTS_VALU", 0) < 1e5:
        continue
    simd = 1024.0
    out[k] = {"launches": n,
              "valu_instructions_per_simd": m["SQ_INSTS_VALU"] / simd, "of_which_mfma": m["SQ_INSTS_MFMA"] / simd,
              "mfma_busy_cycles_per_simd": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, "valu_mfma_coexec_cycles_per_simd": m["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd,
              "active_inst_valu_x4_per_simd": 4.0 * m["SQ_ACTIVE_INST_VALU"] / simd, "busy_cycles_per_se": m["SQ_BUSY_CYCLES"] / 32.0}
json.dump({"command": "tools/pipe_pmc.sh: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra",
           "units": "means per launch; counters summed over the chip by rocprofv3, divided by 1024 SIMDs (instructions, cycles) here", "kernels": out},
          open(f"{R}/gpurun_out/{tag}_pipe_pmc.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_simd"] * kv[1]["launches"]):
    print(f"{k[:48]:48s} n={v['launches']:4d} valu/simd {v['valu_instructions_per_simd']:9.0f} mfma {v['of_which_mfma']:7.0f} mfma_busy {v['mfma_busy_cycles_per_simd']:9.0f} coexec {v['valu_mfma_coexec_cycles_per_simd']:8.0f} active_valu*4 {v['active_inst_valu_x4_per_simd']:9.0f}")
PY
