"""Prints the essentials of a bench.py JSON line:  python tools/show_bench.py gpurun_out/<tag>/bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f %s  ms/step %.4f  kernel avg %.4f ms  frac %.4f  warmup %s (effective %s)  ranks_counted %s" % (
    d["value"], d["unit"], d["ms_per_step"], r.get("kernel_ms_avg", float("nan")), r["frac"], d["warmup"], d.get("warmup_effective"), d.get("ranks_counted")))
print("preconditioning:", d["config"].get("preconditioning"))
cb = d.get("cpu_baseline")
if cb: print("cpu_baseline:", {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb})
for k, v in (d.get("extra") or {}).items():
    if isinstance(v, dict):
        print("  %-22s value %-10s ms %-8s frac %s" % (k, v.get("value"), v.get("ms"), (v.get("roofline") or {}).get("frac")))
