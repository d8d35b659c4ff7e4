#!/bin/bash
# usage (GPU box): tools/pmc_r02.sh > gpurun_out/pmc_r02.txt -- SQ / cache counters of 100-step steady-state rollout launches (float32 and float64 row-major)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dt in float32 float64; do
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
              "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    TS_DTYPE=$dt TS_LAYOUT=row TS_T=100 TS_REPS=4 timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcr_${dt}_$i -- python3 $R/tools/tsweep.py > $R/gpurun_out/pmcr_${dt}_$i.log 2>&1 || echo "pass $i failed"
  done
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for dt in ("float32", "float64"):
    print(f"== k_rollout_pc, row-major {dt}, N = 65536, 100-step launches at the stationary state (last 3 of 4); per launch and per env-wave-step (1024 env waves x 100 steps)")
    for d in sorted(glob.glob(f"gpurun_out/pmcr_{dt}_*/")):
        dur, rows = {}, collections.defaultdict(dict)
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_rollout_pc" in r["Kernel_Name"]:
                    dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_rollout_pc" in r["Kernel_Name"]:
                    rows[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        ids = sorted(dur)[-3:]
        if not ids:
            print("  (no dispatches in", d, ")"); continue
        print("  launch duration under PMC [us]: " + " ".join("%.1f" % dur[i] for i in ids))
        for c, v in sorted(rows.items()):
            m = sum(v[i] for i in ids) / len(ids)
            print("    %-34s %16.0f   per env-wave-step %10.2f" % (c, m, m / (1024 * 100)))
PY
rm -rf gpurun_out/pmcr_*/
