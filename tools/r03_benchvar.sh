#!/bin/bash
# run-to-run spread of the driver's bench command (headline leg = the first leg of a fresh process)
O=gpurun_out/r03f; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg $EXTRA > $O/var_$i.json 2>/dev/null
  python - $O/var_$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
a=d['also']
print('headline %.2f us frac %.3f wall %.3e | feature %.2f | f64 %.2f | step %.2f | steady %.3f' % (d['roofline']['avg_launch_us'], d['roofline']['frac'], d['value'],
      a['rollout_feature']['roofline']['avg_launch_us'], a['rollout_row_float64']['roofline']['avg_launch_us'], a['step_row']['roofline']['avg_launch_us'], d['steady_state']['us_per_step']), flush=True)
PY
done
