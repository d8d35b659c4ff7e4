"""Condense bench.py's JSON line (stdin or a file) into a few readable lines:  python bench.py ... | python tools/bench_line.py <label>"""
import json, sys
label = sys.argv[1] if len(sys.argv) > 1 else ""
src = open(sys.argv[2]) if len(sys.argv) > 2 else sys.stdin
lines = [l for l in src if l.startswith("{")]
if not lines:
    print(f"{label}: NO BENCH LINE")
    sys.exit(1)
d = json.loads(lines[-1])


def leg(name, v):
    r = v["roofline"]
    extra = ""
    if "kernel_only_us" in r:
        extra = f", kernel {r['kernel_only_us']:.2f} us + refresher beside it {r['refresh_us']:.2f} us"
    return (f"  {name:<28} {v['value']:.3e} env-steps/s wall | {r['kernel']} {r['avg_launch_us']:.2f} us per launch of {r['steps_per_launch']:.0f} step(s)"
            f" = {r['achieved']:.0f} GB/s = {r['frac']:.3f} of 8 TB/s ({r['algorithmic_bytes_per_env_step']} B per env-step{extra})")


print(f"## {label}")
print(f"  workload: {d['config']['workload']}; steps {d['steps']}, warm-up {d['warmup']}, n_gpus {d['n_gpus']}")
print(leg("headline " + d["config"]["obs_layout"] + " " + d["config"]["obs_dtype"], d))
if "steady_state" in d:
    s = d["steady_state"]
    print(f"  {'steady_state (400 steps)':<28} {s['us_per_step']:.3f} us per step = {s['achieved']:.0f} GB/s = {s['frac']:.3f}")
for k, v in d.get("also", {}).items():
    print(leg(k, v))
if "episode_boundary" in d and "wall_us" in d["episode_boundary"]:
    b = d["episode_boundary"]
    print(f"  episode_boundary window: {b['wall_us']:.0f} us wall for {b['steps']} steps incl. the terminating one (query {b['finished_query_us']:.0f} us, all-gather {b['all_gather_us']:.0f} us), "
          f"{b['finished_gathered']} episodes gathered, mean return {b['mean_return']:.2f}, mean length {b['mean_length']:.0f}")
if "per_rank" in d:
    print("  per rank: " + "; ".join(f"rank {p['rank']}: wall {p['wall_us']:.1f} us, device {p['device_us']:.1f} us" for p in d["per_rank"]) +
          f"; ranks seen {d.get('rccl_ranks_seen', d.get('ranks_seen'))}")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"  cpu_baseline ({c['kind']}): {c['value']:.3e} env-steps/s on {c['cores']} cores; one thread {c['single_thread']['value']:.3e}")
