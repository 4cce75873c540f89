#!/bin/bash
# Same-box A/B of the whole training loop under environment switches: alternating runs of `bench.py --steps 20 --warmup 3` (no CPU baseline, no
# extras), one line per run.   gpurun -- bash tools/ab_env.sh "BG_X=1" "BG_X=0" [repeats=2]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=${@: -1}
if [[ "$N" =~ ^[0-9]+$ ]]; then set -- "${@:1:$(($#-1))}"; else N=2; fi
for rep in $(seq 1 $N); do
  for v in "$@"; do
    env $v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['phase_ms'].items()}, 'env step us', round(d['roofline_env_step']['avg_launch_us'], 1))"
  done
done
