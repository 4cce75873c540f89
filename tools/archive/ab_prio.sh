for pr in 0 -1 0 -1; do BG_SIDE_PRIORITY=$pr timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_priority=$pr', round(d['value']), {k: round(v,2) for k,v in d['phase_ms'].items()})"; done
