#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch of kernels whose name contains a pattern.
usage: pmc_summary.py <dir> [pattern]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
acc, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-28s %14.1f  (n=%d)" % (k, acc[k] / cnt[k], cnt[k]))
