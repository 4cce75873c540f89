# A/B: weight gradients inside the backward chains (two streams) vs deferred to a joined phase; library vs hand-written kernel
for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; BG_DEFER_WGRAD=$1 BG_FUSED_WGRAD=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$1 hand_wgrad=$2', round(d['value']), {k: round(v,2) for k,v in d['phase_ms'].items()})"; done
