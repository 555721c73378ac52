#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "at::native" in k or "rocclr" in k:
            continue
        short = re.sub(r"^void\s+", "", k).replace("(anonymous namespace)::", "")
        m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", short)
        short = m.group(1) if m else short[:60]
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    n = max(len(v) for v in cs.values())
    print(f"== {k}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):18.1f}")
