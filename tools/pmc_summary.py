#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (pmcN_counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(f"{d}/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
        k = re.sub(r"\(.*", "", k).replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print(f"== {k}  (dispatches per pass: {n})")
    for c in names:
        if c in acc[k]:
            v = acc[k][c]
            print(f"   {c:40s} mean {sum(v)/len(v):16.1f}   min {min(v):14.1f}  max {max(v):14.1f}")
