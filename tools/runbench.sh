for lay in feature row; do for path in step rollout; do timeout -k 10 300 python bench.py --steps 200 --warmup 20 --path $path --obs-layout $lay --no-cpu-baseline $EXTRA > gpurun_out/bench_${path}_${lay}.log 2>&1; python - <<PY
import json
l=[x for x in open("gpurun_out/bench_${path}_${lay}.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$path $lay", "%.3e env-steps/s"%d["value"], "us/step %.2f"%(d["ms_per_step"]*1e3), "GB/s %.0f"%d["roofline"]["achieved"], "frac %.3f"%d["roofline"]["frac"], "launch_us %.2f"%d["roofline"]["avg_launch_us"])
else: print("$path $lay FAILED")
PY
done; done
