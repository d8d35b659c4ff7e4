#!/bin/bash
# ptg_step (k_step_hot) A/B against rl_ptg_amd/lib/exp/libptg_env_head.so: kernel us per launch, eager launches with attached events, 2000 steps
O=gpurun_out/r03c; mkdir -p $O
one() { local lib=$1; shift
  if [ "$lib" = "head" ]; then export PTG_LIB_PATH=$PWD/rl_ptg_amd/lib/exp/libptg_env_head.so; else unset PTG_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also --path step --launch eager "$@" > $O/tmp.json 2>/dev/null
  python - "$lib $*" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/r03c/tmp.json') if l.startswith('{')][-1])
except Exception:
    print(sys.argv[1], 'FAILED'); sys.exit(0)
r = d['roofline']
print('%-60s kernel us/launch %.3f frac %.3f' % (sys.argv[1], r['avg_launch_us'], r['frac']), flush=True)
PY
}
for rep in 1 2 3; do for lib in head new; do one $lib --envs 65536; one $lib --envs 65536 --out-dtype float64; one $lib --envs 16384; done; done
