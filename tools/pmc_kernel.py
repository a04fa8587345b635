#!/usr/bin/env python3
"""Average the counters of rocprofv3 --pmc passes per kernel whose name contains a substring.

    python tools/pmc_kernel.py <dir> <substring>
"""
import csv, glob, os, sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if sub in row["Kernel_Name"]:
                acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
for name, per in sorted(acc.items()):
    v = list(per.values())
    print(f"{name:32s} {sum(v) / len(v):16.1f}  ({len(v)} dispatches)")
