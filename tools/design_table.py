"""Prints DESIGN.md section 5's table rows from a bench.py output line: python tools/design_table.py profiles/<tag>_bench.json"""
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def fmt(name, v, unit, ms, r):
    tr = "n/a" if r["traffic"] is None else f"{r['traffic']/1e6:.1f} / {r['algorithmic_bytes']/1e6:.1f} MB"
    ach = f"{r['achieved']} {r['unit']}"
    print(f"| {name} | {v:.3g} {unit} ({ms:.4g} ms) | {r['bound']}: {ach} | {r['frac']} | {r.get('valu_issue_frac')} | {tr} |")
fmt("**headline** " + b["config"]["workload"][:60], b["value"], b["unit"], b["ms_per_step"], b["roofline"])
for a in b.get("also", []):
    fmt(a["workload"][:75], a["value"], a["unit"], a["ms_per_step"], a["roofline"])
print("cpu:", b.get("cpu_baseline"))
r = b["roofline"]
print("headline: executed flop/work unit", r["executed_flops_per_work_unit"], "work units/env-step", r["work_units_per_env_step"], "work_equiv", r["work_equiv"]["tflops"])
