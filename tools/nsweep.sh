#!/bin/bash
# usage (GPU box): tools/nsweep.sh > gpurun_out/nsweep.txt   -- bench.py over batch sizes / layouts / dtypes, one line each
one() { timeout -k 10 400 python bench.py --no-cpu-baseline --no-boundary-leg "$@" > gpurun_out/bench_sweep.log 2>&1; python - "$@" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/bench_sweep.log') if l.startswith('{')][-1])
except Exception as e:
    print(' '.join(sys.argv[1:]), 'FAILED'); sys.exit(0)
r = d['roofline']
msg = '%-58s %-8s dev us/step %8.3f  %7.0f GB/s frac %.3f  wall env-steps/s %.3e' % (' '.join(sys.argv[1:]), 'rollout' if 'rollout' in d['config']['path'] else 'step', r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['achieved'], r['frac'], d['value'])
s = d.get('steady_state')
if s:
    msg += ' | steady us/step %.3f frac %.3f' % (s['us_per_step'], s['frac'])
print(msg, flush=True)
PY
}
# one path per process (--no-also): above 131 072 envs a second leg in the same process shows a one-off stall between the last
# kernel and the closing event, which would be charged to it
for n in 4096 16384 65536 131072 262144; do one --envs $n --no-also; one --envs $n --path step --no-also; done
one --envs 1048576 --steps 200 --warmup 200 --no-also
one --envs 1048576 --steps 200 --warmup 200 --path step --no-also
one --envs 65536 --obs-layout feature --no-also
one --envs 65536 --obs-layout sb3_flat --no-also
one --envs 65536 --out-dtype float64 --no-also
one --envs 65536 --out-dtype float64 --obs-layout feature --no-also
one --envs 65536 --out-dtype float64 --obs-layout sb3_flat --no-also
one --envs 65536 --obs-layout split --no-also
one --envs 65536 --out-dtype float64 --obs-layout split --no-also
one --envs 65536 --out-dtype float64 --path step --no-also
one --envs 65536 --path step --launch eager --no-also
one --envs 65536 --noise tape --no-also
one --envs 100000 --no-also
