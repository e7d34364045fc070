#!/usr/bin/env python3
"""A few lines out of a bench.py JSON line (development aid): bench_summary.py <file with the line>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"headline {d['value']:.1f} steps/s, {d['ms_per_step']:.2f} ms per pass; {r['kernel']} {r['kernel_ms']:.2f} ms, "
      f"{r['achieved']:.0f} GB/s = {r['frac']:.4f} of the HBM peak, traffic {r.get('traffic')}")
for k in ("warm", "cold"):
    if k in d:
        print(f"  {k:22s} {d[k]['value']:10.1f} steps/s  {d[k]['ms_per_pass']:9.2f} ms")
for k, v in d.get("configs", {}).items():
    print(f"  {k:22s} {v['value']:10.1f} steps/s  {v['ms_per_pass']:9.2f} ms  {v['roofline']['kernel']}  frac {v['roofline']['frac']:.4f}"
          f"  traffic {v['roofline'].get('traffic')}" + (f"  cpu {v['cpu_baseline']['value']:.0f}" if "cpu_baseline" in v else ""))
if "throughput_mode" in d:
    print("  throughput mode       ", round(d["throughput_mode"]["value"], 1), "steps/s")
if "cpu_baseline" in d:
    print("  cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")})
if "strong" in d:
    for k, v in d["strong"].items():
        print(f"  strong {k}: {v['value']:.1f} steps/s, slowest instance {v['bound']['slowest_instance_in_kernel_ms']:.1f} ms of {v['ms_per_step']:.1f} ms")
