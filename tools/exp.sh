for f in 0 1 2 3; do
PTG_DEBUG_FLAGS=$f timeout -k 10 300 python bench.py --steps 200 --warmup 20 --path rollout --obs-layout feature --no-cpu-baseline > gpurun_out/exp_$f.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/exp_$f.log") if x.startswith("{")]
d=json.loads(l[-1]); print("dbg=$f", "us/step %.2f"%(d["roofline"]["avg_launch_us"]/200))
PY
done
