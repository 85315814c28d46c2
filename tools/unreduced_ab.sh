for k in 1; do
PTTS_BRANCH_STREAM=$k python3 bench.py --no-variants --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline --no-roofline --steps 60 --warmup 15 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('branch_stream=$k headline %.3f M; unreduced %.3f M (%.2f ms)' % (j['value']/1e6, j['all_exact_work_reductions_off']['value']/1e6, j['all_exact_work_reductions_off']['ms_per_step']))
"
done
