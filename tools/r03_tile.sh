#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_rollout_parity.py tests/test_batch_edges.py tests/test_hip_parity.py tests/test_split_layout.py tests/test_vec_env_host.py -m gpu -q -x > $O/tests_tile.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_tile.log
one() { local lib=$1; shift
  if [ "$lib" = "head" ]; then export PTG_LIB_PATH=$PWD/rl_ptg_amd/lib/exp/libptg_env_head.so; else unset PTG_LIB_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary-leg --no-also "$@" > $O/tmp.json 2>/dev/null
  python - "$lib $*" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open('gpurun_out/r03c/tmp.json') if l.startswith('{')][-1])
except Exception:
    print(sys.argv[1], 'FAILED'); sys.exit(0)
r = d['roofline']; s = d.get('steady_state') or {}
print('%-72s dev us/step %.3f frac %.3f | steady %.3f' % (sys.argv[1], r['avg_launch_us'] * r['launches_timed'] / d['steps'], r['frac'], s.get('us_per_step', 0)), flush=True)
PY
}
for rep in 1 2; do
for lib in head new; do
  one $lib --envs 4096 --scenario 2 --operation OP2
  one $lib --envs 65536 --obs-layout split
  one $lib --envs 65536 --obs-layout sb3_flat
  one $lib --envs 65536 --steps 20 --warmup 5
  one $lib --envs 65536 --path step --launch eager
done
done
true
