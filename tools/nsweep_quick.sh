#!/bin/bash
# a few lines of tools/nsweep.sh (layouts, ragged batch) for quick A/B runs
. /dev/null
one() { timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_sweep.log 2>&1; python - "$@" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/bench_sweep.log') if l.startswith('{')][-1])
except Exception as e:
    print(' '.join(sys.argv[1:]), 'FAILED'); sys.exit(0)
r = d['roofline']
msg = '%-58s dev us/step %8.3f  %7.0f GB/s frac %.3f' % (' '.join(sys.argv[1:]), r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['achieved'], r['frac'])
s = d.get('steady_state')
if s:
    msg += ' | steady us/step %.3f frac %.3f' % (s['us_per_step'], s['frac'])
print(msg, flush=True)
PY
}
one --envs 65536 --no-also
one --envs 65536 --obs-layout sb3_flat --no-also
one --envs 100000 --no-also
one --envs 100000 --obs-layout feature --no-also
one --envs 65536 --out-dtype float64 --no-also
one --envs 65000 --path step --no-also
