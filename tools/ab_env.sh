#!/bin/bash
# Same-box A/B of bench.py under two environment settings, alternating, REPS times each:
#   bash tools/ab_env.sh "BG_DEFER_FINISH=1" "BG_DEFER_FINISH=0" [reps]      (AB_ARGS="--steps 30": extra bench.py arguments)
# prints value / ms per iteration / phase split of every run (bench.py --no-cpu-baseline --no-extra).
A="$1"; B="$2"; REPS=${3:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for i in $(seq 1 $REPS); do
  for cfg in "$A" "$B"; do
    env $cfg python3 $R/bench.py --no-cpu-baseline --no-extra ${AB_ARGS:-} > $R/gpurun_out/ab_tmp.json 2> $R/gpurun_out/ab_tmp.err
    python3 - "$cfg" <<'PY'
import json, sys
d = json.loads([l for l in open(__import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/ab_tmp.json") if l.startswith("{")][-1])
print(f"{sys.argv[1]:28s} {d['value']/1e6:6.3f} M env-steps/s  {d['ms_per_step']:7.3f} ms  rollout {d['phase_ms']['rollout']:.3f}  update {d['phase_ms']['update']:.3f}", flush=True)
PY
  done
done
