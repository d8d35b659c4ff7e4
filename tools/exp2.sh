run() { # name, env, args
  env $2 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $3 > gpurun_out/exp_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "value %.3e"%d["value"], "wall us/step %.2f"%(d["ms_per_step"]*1e3), "dev us/launch %.2f"%d["roofline"]["avg_launch_us"], "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp_$1.log").read()[-800:])
PY
}
run roll_pipe "A=1" "--path rollout"
run roll_nopipe "PTG_NO_PIPELINE=1" "--path rollout"
run roll_pipe_nostore "PTG_DEBUG_FLAGS=1" "--path rollout"
run roll_generic "PTG_NO_FAST_KERNELS=1" "--path rollout"
run step_fast "A=1" "--path step"
run roll_262k "A=1" "--path rollout --envs 262144 --steps 100"
run step_262k "A=1" "--path step --envs 262144 --steps 100"
