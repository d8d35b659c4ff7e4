run() { # name env args
  env $2 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-also $3 > gpurun_out/exp10_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp10_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "value %.3e"%d["value"], "dev us/launch %.2f"%(d["roofline"]["avg_launch_us"]), "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp10_$1.log").read()[-600:])
PY
}
run roll A=1 "--path rollout"
run roll_generic PTG_NO_HOT_KERNELS=1 "--path rollout"
run roll_f64 A=1 "--path rollout --out-dtype float64"
run step_f64 A=1 "--path step --out-dtype float64"
run step_generic PTG_NO_HOT_KERNELS=1 "--path step"
run step_eager A=1 "--path step --launch eager"
