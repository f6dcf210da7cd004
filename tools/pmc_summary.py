#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    sys.exit("no counter_collection.csv under " + d)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void dtf::", "").replace("dtf::", "")[:44]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in acc.values() for c in v})
print(f"{'kernel':44s} " + " ".join(f"{c[-18:]:>18s}" for c in counters))
for name, cs in sorted(acc.items()):
    if not name.startswith(("conv", "gemm", "head", "reduce", "sgd", "_ZN3dtf", "transpose", "sample", "adam", "fused")):
        continue
    print(f"{name:44s} " + " ".join(f"{(sum(cs[c])/len(cs[c]) if c in cs else 0):18.0f}" for c in counters))
