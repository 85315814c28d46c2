#!/bin/bash
# A/B on the GPU box: generator step with the BLSTM recurrences as hipGraph replays (PTTS_LSTM_GRAPH) and with the side branch's
# nodes created last (PTTS_SIDE_DEFER: its backward chain is then enqueued first).  Prints value / critic / generator ms per setting.
LEGS="--no-variants --no-unreduced --no-host-leg --no-reference-shape --no-bf16-leg --no-gated-leg --no-cpu-baseline"
for dt in f32 bf16; do
for g in 0 1; do for d in 0 1; do
  PTTS_LSTM_GRAPH=$g PTTS_SIDE_DEFER=$d python3 bench.py --dtype $dt $LEGS --steps 36 --warmup 12 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$dt graph=$g defer=$d value %.3f M  ms/step %.3f  critic %.3f  gen %.3f' % (j['value']/1e6, j['ms_per_step'], j.get('critic_step_ms',0), j.get('generator_step_ms',0)), j.get('lstm_graph'))
"
done; done; done
