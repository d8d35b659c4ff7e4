#!/bin/bash
# bench lines at every BASELINE.json config on its own scenario -> gpurun_out/r03/configs.txt
O=gpurun_out/r03; mkdir -p $O; OUT=$O/configs.txt; : > $OUT
b() { local label=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/cfg_tmp.json 2> $O/cfg_tmp.err; local rc=$?
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $label" >> $OUT; exit 1; fi
      python tools/bench_line.py "$label" $O/cfg_tmp.json >> $OUT 2>&1 || tail -5 $O/cfg_tmp.err >> $OUT; echo >> $OUT; }
b "configs[1]: N = 4096, BS2/OP2 (default run: 400 + 2000 steps)" --scenario 2 --operation OP2 --envs 4096 --no-cpu-baseline
b "configs[1]: N = 4096, BS2/OP2, driver window (5 + 20 steps)" --scenario 2 --operation OP2 --envs 4096 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg
b "configs[2]: N = 65536, BS1/OP1 (default run: 400 + 2000 steps)" --no-cpu-baseline
b "configs[2]: N = 65536, BS1/OP1, driver window (5 + 20 steps)" --steps 20 --warmup 5
b "configs[3]: N = 262144, BS3/OP2 (100 + 400 steps)" --scenario 3 --operation OP2 --envs 262144 --steps 400 --warmup 100 --no-cpu-baseline
b "configs[3]: N = 262144, BS3/OP2, driver window (5 + 20 steps)" --scenario 3 --operation OP2 --envs 262144 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg
b "configs[4] per-rank leg: N = 65536 of 524288, BS1+2+3 mixed, OP2 (400 + 2000 steps)" --mixed-scenarios --operation OP2 --no-cpu-baseline
b "configs[4] per-rank leg, driver window (5 + 20 steps)" --mixed-scenarios --operation OP2 --steps 20 --warmup 5 --no-cpu-baseline --no-boundary-leg
cat $OUT
