# A/B: fp32-MFMA layer kernels vs the split-bf16 form (9 and 6 products) inside the full iteration
for s in 0 9 6; do BG_GEMM_SPLIT=$s timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$s', round(d['value']), {k: round(v,2) for k,v in d['phase_ms'].items()})"; done
