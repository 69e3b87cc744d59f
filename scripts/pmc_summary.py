#!/usr/bin/env python
"""Aggregates rocprofv3 counter_collection CSVs into a per-kernel summary (mean counter value per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = name.replace("(anonymous namespace)::", "").split("(")[0][-90:]
            out[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "sum"])
for k in sorted(out):
    for c, vals in sorted(out[k].items()):
        w.writerow([k, c, len(vals), sum(vals) / len(vals), sum(vals)])
