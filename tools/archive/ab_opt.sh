# A/B of the fused mini-epoch tail (bg_optimizer_step) against the separate launches
for f in 1 0 1 0; do BG_FUSED_OPT=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_opt=$f', round(d['value']), {k: round(v,2) for k,v in d['phase_ms'].items()}, 'update frac', round(d['roofline_update']['frac'],3))"; done
