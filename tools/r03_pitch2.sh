#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for pad in 0 256 512 1024 4096 8448 34816 65536 133120; do
  PTG_FM_PAD_BYTES=$pad timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also --out-dtype float64 --obs-layout feature > $O/pitch_tmp.json 2>/dev/null
  python - "pad $pad B" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r03/pitch_tmp.json') if l.startswith('{')][-1])
r = d['roofline']; s = d.get('steady_state') or {}
print('%-16s pitch %s  dev us/step %.3f frac %.3f | steady %.3f frac %.3f' % (sys.argv[1], d['config'].get('obs_plane_pitch'), r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['frac'], s.get('us_per_step', 0), s.get('frac', 0)), flush=True)
PY
done
for n in 61440 65280 65536; do
  PTG_FM_PAD_BYTES=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also --out-dtype float64 --obs-layout feature --envs $n > $O/pitch_tmp.json 2>/dev/null
  python - "N $n no pad" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r03/pitch_tmp.json') if l.startswith('{')][-1])
r = d['roofline']; s = d.get('steady_state') or {}
print('%-16s pitch %s  dev us/step %.3f frac %.3f | steady %.3f frac %.3f' % (sys.argv[1], d['config'].get('obs_plane_pitch'), r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['frac'], s.get('us_per_step', 0), s.get('frac', 0)), flush=True)
PY
done
