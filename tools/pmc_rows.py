#!/usr/bin/env python3
"""tools/pmc_rows.py <counter_collection.csv> — per kernel: mean of every counter over the kernel's dispatches (rocprofv3 --pmc CSV), one line per (kernel, counter)."""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])): acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()): print("%-28s %-30s %16.0f  (%d dispatches)" % (k[:28], c, sum(v) / len(v), len(v)))
