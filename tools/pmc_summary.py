#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes: per kernel, mean of each counter over dispatches."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(root.rglob("*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")[:70]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "gemm" not in k and "attention" not in k and "resample" not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
