#!/bin/bash
# round-3 batch 1: GPU tests of the new refresher code, refresher modes side by side, the persistent-step and PCIe probes, one bench line
O=gpurun_out/r03; mkdir -p $O
run() {   # run <seconds> <log> <cmd...>: a step that times out ends the batch (no further GPU step after a kill)
    local lim=$1 log=$2; shift 2
    timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?
    echo "rc=$rc $*" >> $O/batch1.status
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi
}
: > $O/batch1.status
run 600 $O/tests1.log python -m pytest tests -m gpu -x -q
run 200 $O/ab_default.txt python tools/refresh_ab.py
PTG_REFRESH_MODE=legacy run 200 $O/ab_legacy.txt python tools/refresh_ab.py
PTG_REFRESH_MODE=head run 200 $O/ab_head.txt python tools/refresh_ab.py
PTG_NO_REFRESH=1 run 200 $O/ab_none.txt python tools/refresh_ab.py
TS_DTYPE=float64 run 200 $O/ab_default_f64.txt python tools/refresh_ab.py
TS_DTYPE=float64 PTG_REFRESH_MODE=legacy run 200 $O/ab_legacy_f64.txt python tools/refresh_ab.py
run 120 $O/persist.txt tools/bin/persist_probe 2000
run 120 $O/pcie.txt tools/bin/pcie_probe
run 300 $O/bench20.json python bench.py --gpus 1 --steps 20 --warmup 5
cat $O/batch1.status
tail -3 $O/tests1.log
