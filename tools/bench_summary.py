"""One-line digest of a bench.py JSON line (headline, side legs, tier shares): python tools/bench_summary.py <bench.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]
print("headline %.3f M (%.2f ms) | scale 0.3 / 0.05: %s | policy %.3f M (bigger-tier share %.2f) | config2 %.2f M | config4 %.2f M (%s launches) | bigger-tier share %.2f | roofline frac %.3g | cpu %s" % (
    d["value"] / 1e6, d["ms_per_step"], {k: round(v / 1e6, 3) for k, v in c.get("small_action_env_steps_per_s", {}).items()},
    c.get("policy_driven", {}).get("value", 0) / 1e6, c.get("policy_driven", {}).get("heavy_tier_fraction", -1), (c.get("config2_env_steps_per_s") or 0) / 1e6,
    (c.get("config4_env_steps_per_s") or 0) / 1e6, (c.get("config_legs", {}).get("config4") or {}).get("launches_per_step"), c.get("heavy_tier_fraction", -1),
    d["roofline"]["frac"], {k: d.get("cpu_baseline", {}).get(k) for k in ("value", "cores", "parallel_efficiency")}))
