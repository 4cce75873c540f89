for cfg in "0 256" "1 256" "1 128"; do set -- $cfg; BG_FUSED_WGRAD=$1 BG_WGRAD_WORKGROUPS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad=$1 wgs=$2', round(d['value']), d['phase_ms'], round(d['roofline_update']['frac'],3))"; done
