"""Condense the rocprofv3 outputs of tools/prof_round.sh into profiles/<tag>_*.{csv,md,json} (the files that are committed)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
os.makedirs("profiles", exist_ok=True)
ours = ("k_step", "k_rollout", "k_reset", "k_build", "k_fill", "k_init", "k_zero", "k_extract")


def short(name):
    for k in ("k_step_hot", "k_rollout_pc", "k_extract_keys", "k_step", "k_rollout", "k_reset", "k_build_records", "k_build_argmin", "k_build_fast", "k_fill_noise", "k_init_state", "k_zero_noise_count"):
        if k in name:
            return k
    return name[:60]


rows = []
for f in glob.glob(f"gpurun_out/prof_{tag}_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        nm = r["Name"]
        w.writerow([short(nm) if any(k in nm for k in ours) else nm[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

trace = defaultdict(list)
meta = {}
for f in glob.glob(f"gpurun_out/prof_{tag}_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        trace[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}

pmc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/prof_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))

lines = [f"# rocprofv3 summary, {tag}", "",
         "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (defaults: 400 warm-up + 2000 timed steps of both",
         "paths: the k_rollout_pc dispatches are the 400-step warm-up launch and the timed rollout's 401 + 401 + 401 + 401 + 396-step launches",
         "(400 steps per launch on average, as in bench.py's avg_launch_us); every k_step_hot dispatch is one vector step; 1x MI355X, N = 65536 envs,",
         "BS1/OP1, float32 feature-major obs, in-kernel RNG).  HBM counters: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of",
         "`bench.py --steps 100 --warmup 10 --launch eager`.  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed reads);",
         "counter unit KiB.", "",
         "| kernel | dispatches | avg us | min us | max us | VGPR | SGPR | LDS B | block | grid |", "|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(trace, key=lambda x: -sum(v[1] for v in trace[x])):
    if not any(k.startswith(o) for o in ours):
        continue
    d, m = [x[1] for x in sorted(trace[k])], meta[k]
    lines.append(f"| {k} | {len(d)} | {sum(d) / len(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | {max(d) / 1e3:.2f} | {m['VGPR_Count']} | {m['SGPR_Count']} | {m['LDS_Block_Size']} | {m['Workgroup_Size_X']} | {m['Grid_Size_X']} |")
    if k == "k_rollout_pc" and len(d) > 1:
        timed = d[1:]                                     # trace order: the first dispatch is the warm-up launch from reset
        note = (f"`k_rollout_pc`: the {len(timed)} dispatches of the timed rollout average {sum(timed) / len(timed) / 1e3:.2f} us "
                f"(bench.py's `roofline.avg_launch_us`, same launches without the profiler: 606-620 us); the warm-up launch from reset took {d[0] / 1e3:.2f} us.")
traffic = {}
if "note" in dir():
    lines += ["", note]
lines += ["", "| kernel | FETCH_SIZE KiB/launch (raw) | read bytes/launch (x2 corrected) | WRITE_SIZE KiB/launch | HBM bytes/launch |", "|---|---|---|---|---|"]
for k in ("k_step_hot", "k_rollout_pc"):
    if k in pmc:
        fs = pmc[k].get("FETCH_SIZE", [])
        ws = pmc[k].get("WRITE_SIZE", [])
        # rollout: the 10-step warm-up launch and the 100-step launch differ; use the largest (the timed launch)
        f = max(fs) if fs else None
        w_ = max(ws) if ws else None
        tot = (2 * f * 1024 if f else 0) + (w_ * 1024 if w_ else 0)
        lines.append(f"| {k} | {f} | {2 * f * 1024 if f else None} | {w_} | {tot} |")
        traffic[("step" if k == "k_step_hot" else "rollout") + "_bytes_per_launch"] = tot
        traffic[("step" if k == "k_step_hot" else "rollout") + "_pmc_launch_steps"] = 1 if k == "k_step_hot" else 100
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
if "rollout_bytes_per_launch" in traffic and "step_bytes_per_launch" in traffic:      # what bench.py reports as roofline.traffic
    json.dump({"source": f"profiles/{tag}_summary.md", "step_bytes_per_launch": traffic["step_bytes_per_launch"],
               "rollout_bytes_per_step": traffic["rollout_bytes_per_launch"] / traffic["rollout_pmc_launch_steps"]},
              open("profiles/traffic_latest.json", "w"), indent=1)
print("\n".join(lines))
