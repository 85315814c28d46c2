#!/bin/bash
# A/B on one box: the critic's context branch on a side stream beside the spectral stacks (cfg.train_wgan_ctx_stream)
cd "$GRAFT_REPO_ROOT"
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline --no-roofline"
for i in 1 2; do
for v in 0 1; do
PTTS_CTX_STREAM=$v python3 bench.py $LEGS --steps 60 --warmup 15 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ctx_stream=$v value %.3f M ms/step %.3f critic %.3f gen %.3f' % (j['value']/1e6, j['ms_per_step'], j['critic_step_ms'], j['generator_step_ms']), j['config']['hipgraph'].get('tuning_ms'))
"
done
done
