run() { # name env args
  env $2 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $3 > gpurun_out/exp5_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp5_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "value %.3e"%d["value"], "dev us/step %.2f"%(d["roofline"]["avg_launch_us"]/200), "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp5_$1.log").read()[-600:])
PY
}
run lpw64 "PTG_LPW=64" "--path rollout"
run lpw32 "PTG_LPW=32" "--path rollout"
run lpw16 "PTG_LPW=16" "--path rollout"
run lpw32_nopipe "PTG_LPW=32 PTG_NO_PIPELINE=1" "--path rollout"
run lpw16_nopipe "PTG_LPW=16 PTG_NO_PIPELINE=1" "--path rollout"
run lpw32_b128 "PTG_LPW=32 PTG_BLOCK=128" "--path rollout"
