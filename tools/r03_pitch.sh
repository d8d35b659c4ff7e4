#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_batch_edges.py -m gpu -q -k "pitch or boundaries" 2>&1 | tail -6
for args in "--out-dtype float64 --obs-layout feature" "--obs-layout feature" "--out-dtype float64 --obs-layout feature --steps 20 --warmup 5"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also $args > $O/pitch_tmp.json 2>/dev/null
  python - "$args" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r03/pitch_tmp.json') if l.startswith('{')][-1])
r = d['roofline']; s = d.get('steady_state') or {}
print('%-70s pitch %s  dev us/step %.3f frac %.3f | steady %.3f frac %.3f' % (sys.argv[1], d['config'].get('obs_plane_pitch'), r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['frac'], s.get('us_per_step', 0), s.get('frac', 0)), flush=True)
PY
done
