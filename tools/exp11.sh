run() { # name env args
  env $2 timeout -k 10 300 python bench.py --warmup 10 --no-cpu-baseline --no-also $3 > gpurun_out/exp11_$1.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp11_$1.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$1", "value %.3e"%d["value"], "dev us/launch %.2f"%(d["roofline"]["avg_launch_us"]), "GB/s %.0f"%d["roofline"]["achieved"], "frac %.3f"%d["roofline"]["frac"])
else: print("$1 FAILED"); print(open("gpurun_out/exp11_$1.log").read()[-600:])
PY
}
for n in 4096 16384 65536 131072 262144 524288 1048576; do
  st=$((13107200 / n)); if [ $st -gt 400 ]; then st=400; fi; if [ $st -lt 20 ]; then st=20; fi
  run roll_$n A=1 "--path rollout --envs $n --steps $st"
done
run roll_262k_pipe PTG_PIPE=1 "--path rollout --envs 262144 --steps 50"
run roll_65k_nopipe PTG_PIPE=0 "--path rollout --envs 65536 --steps 200"
for n in 4096 65536 262144 1048576; do
  st=$((13107200 / n)); if [ $st -gt 400 ]; then st=400; fi; if [ $st -lt 20 ]; then st=20; fi
  run step_$n A=1 "--path step --envs $n --steps $st"
done
