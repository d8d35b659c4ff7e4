#!/bin/bash
# upper bound of key run-ahead / speculation for chain-bound shapes: product library vs the -DPTG_ABLATE_KEYLAG build (timing only, wrong results)
O=gpurun_out/r03; mkdir -p $O
one() { local lib=$1; shift
  if [ "$lib" = "ablate" ]; then export PTG_LIB_PATH=$PWD/rl_ptg_amd/lib/exp/libptg_env_keylag.so; else unset PTG_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also "$@" > $O/keylag_tmp.json 2>/dev/null
  python - "$lib $*" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/r03/keylag_tmp.json') if l.startswith('{')][-1])
except Exception:
    print(sys.argv[1], 'FAILED'); sys.exit(0)
r = d['roofline']; s = d.get('steady_state') or {}
print('%-60s dev us/step %.3f frac %.3f | steady %.3f' % (sys.argv[1], r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['frac'], s.get('us_per_step', 0)), flush=True)
PY
}
for lib in product ablate; do
  one $lib --envs 4096 --scenario 2 --operation OP2
  one $lib --envs 16384
  one $lib --envs 65536
  one $lib --envs 65536 --obs-layout split
  one $lib --envs 65536 --obs-layout feature
done
