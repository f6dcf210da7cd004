#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel µs per step."""
import csv
import glob
import sys

d = sys.argv[1]
calls_per_kernel = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + d)
rows = list(csv.DictReader(open(files[0])))
tot = 0.0
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>9s} {'us/step':>9s}")
for r in rows:
    name = r["Name"].replace("void dtf::", "").replace("dtf::", "")[:60]
    calls = int(r["Calls"])
    if calls < calls_per_kernel:
        continue
    per_step = float(r["TotalDurationNs"]) / 1e3 / calls_per_kernel
    tot += per_step
    print(f"{name:60s} {calls:6d} {float(r['AverageNs'])/1e3:9.2f} {per_step:9.2f}")
print(f"{'TOTAL kernel time per step':60s} {'':6s} {'':9s} {tot:9.2f}")
