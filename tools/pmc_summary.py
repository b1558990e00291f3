#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV passes: per kernel (short name), mean counter value per dispatch."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            m = re.search(r"(k_\w+)(<[^>]*>)?", name)
            if not m:
                continue
            short = m.group(1) + (m.group(2) or "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
