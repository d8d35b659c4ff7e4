run() { # name env args
  env $2 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-also $3 > gpurun_out/exp9_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp9_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "value %.3e"%d["value"], "dev us/launch %.2f"%(d["roofline"]["avg_launch_us"]), "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp9_$1.log").read()[-600:])
PY
}
run roll_pc A=1 "--path rollout"
run roll_nopc PTG_NO_PC=1 "--path rollout"
run step A=1 "--path step"
