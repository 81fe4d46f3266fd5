#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from *_counter_collection.csv files.
usage: python scripts/pmc_summary.py <csv> [<csv> ...]   -> prints kernel, counter, mean per dispatch, dispatches"""
import collections
import csv
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][:60]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, ctr), vals in sorted(acc.items()):
        if "at::native" in name or "rocclr" in name:
            continue
        print(f"{name:62s} {ctr:24s} mean {sum(vals) / len(vals):16.1f}  n={len(vals)}")
