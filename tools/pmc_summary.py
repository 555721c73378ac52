#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "at::native" in k or "rocclr" in k:
            continue
        short = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    n = max(len(v) for v in cs.values())
    print(f"== {k}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):18.1f}")
