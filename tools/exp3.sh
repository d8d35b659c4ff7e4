run() { env PTG_DEBUG_FLAGS=$1 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --path rollout > gpurun_out/exp3_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp3_$1.log") if x.startswith("{")]
d=json.loads(l[-1]); print("dbg=$1", "us/step %.2f"%(d["roofline"]["avg_launch_us"]/200))
PY
}
for f in 0 1 2 4 8 3 7 15; do run $f; done
