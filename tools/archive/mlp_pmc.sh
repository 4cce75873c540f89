#!/bin/bash
# SQ counters of the MFMA layer kernels alone on the GPU (tools/mlp_probe.py): where do the cycles of a wave go?
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_mlp_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/tools/mlp_probe.py > $OUT.a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $R/tools/mlp_probe.py > $OUT.b.log 2>&1 || echo "pass b failed (counter names)"
python3 - <<'PY'
import collections, csv, glob, json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/prof_mlp_pmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "mlp_fwd" in n:
            acc[n + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open(f"{R}/gpurun_out/mlp_pmc.json", "w"), indent=1)
for k, cs in sorted(out.items()):
    print(k, {c: round(v) for c, v in cs.items()})
PY
