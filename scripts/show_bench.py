"""Prints the few numbers of a bench.py JSON line one looks at while iterating: value, per-stage totals, secondary blocks.
usage: python scripts/show_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"], 1), "it/s  ms/step", round(d["ms_per_step"], 4))
st = d.get("stages", {})
print("stage totals ms:", {k: round(v, 3) for k, v in st.get("ms_total", {}).items()})
ps = st.get("per_stage", {})
print("per launch ms:", {k: round(v.get("avg_ms", 0), 4) for k, v in ps.items() if isinstance(v, dict)})
print("roofline:", d.get("roofline"))
for k in ("value_including_setup", "value_after_outlier_removal", "recent_window", "secondary"):
    if k in d: print(k, json.dumps(d[k])[:600])
print("config:", {k: d["config"].get(k) for k in ("reduced_system_solves", "trials_served_speculatively", "persist_fallbacks")})
