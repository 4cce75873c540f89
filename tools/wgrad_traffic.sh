#!/bin/bash
# HBM-side traffic and duration of mlp_wgrad_group_kernel inside the bench loop (rocprofv3: one --stats pass, one FETCH_SIZE pass, one WRITE_SIZE pass).
#   gpurun -- bash tools/wgrad_traffic.sh <tag>      [BG_LIB=... to profile another library build]
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_wg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT.trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT.f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT.w.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys
root, tag = sys.argv[1], sys.argv[2]
st = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(st)):
    if r["Name"].startswith("mlp_wgrad_group_kernel"):
        print(tag, "mlp_wgrad_group_kernel avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
tot = {}
for p in ("fetch", "write"):
    vals = []
    for f in glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("mlp_wgrad_group_kernel"):
                vals.append(float(r["Counter_Value"]))
    tot[p] = sum(vals) / max(len(vals), 1)
print(tag, "FETCH_SIZE KiB", tot["fetch"], "WRITE_SIZE KiB", tot["write"], "hbm_bytes (2 x fetch + write, 16-byte-per-lane loads)", (2 * tot["fetch"] + tot["write"]) * 1024)
PY
