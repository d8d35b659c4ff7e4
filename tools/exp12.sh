run() { # name env args
  env $2 timeout -k 10 300 python bench.py --warmup 10 --no-cpu-baseline --no-also $3 > gpurun_out/exp12_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp12_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "dev us/launch %.2f"%(d["roofline"]["avg_launch_us"]), "GB/s %.0f"%d["roofline"]["achieved"], "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp12_$1.log").read()[-600:])
PY
}
run roll65k A=1 "--path rollout --envs 65536 --steps 200"
run roll65k_blocked PTG_DEBUG_FLAGS=32 "--path rollout --envs 65536 --steps 200"
run roll1m A=1 "--path rollout --envs 1048576 --steps 20"
run roll1m_blocked PTG_DEBUG_FLAGS=32 "--path rollout --envs 1048576 --steps 20"
run step1m A=1 "--path step --envs 1048576 --steps 20"
run step1m_blocked PTG_DEBUG_FLAGS=32 "--path step --envs 1048576 --steps 20"
run step65k_blocked PTG_DEBUG_FLAGS=32 "--path step --envs 65536 --steps 200"
