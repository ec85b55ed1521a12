#!/usr/bin/env python3
"""Summarise one tools/profile_round.sh directory: kernel-trace stats + FETCH_SIZE / WRITE_SIZE per launch."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
out = {}
for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (%s)" % f.split("/")[-1])
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        short = name[:90]
        print("%-92s calls %6s avg %10.1f ns  min %8s max %8s  %6s%%" % (short, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
        if "::step_" in name:
            out["step_kernel_avg_ns"] = float(r["AverageNs"])
            out["step_kernel_calls"] = int(r["Calls"])
            out["step_kernel_name"] = name
for key in ("fetch", "write"):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/%s/**/*counter_collection.csv" % key, recursive=True):
        for r in csv.DictReader(open(f)):
            if "::step_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
    for k in acc:
        v = acc[k] / cnt[k]
        print("== %s per step_kernel launch: %.1f (counter units; x1024 = bytes -> %.3f MB)  n=%d" % (k, v, v * 1024 / 1e6, cnt[k]))
        out[k + "_per_launch_raw"] = v
print(json.dumps(out))
