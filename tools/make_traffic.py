#!/usr/bin/env python3
"""profiles/traffic.json from the summaries tools/profile_round.sh wrote: per configuration the rocprofv3 kernel average and the
FETCH_SIZE / WRITE_SIZE per launch (separate --pmc passes), tagged with the sha256 of the library that was profiled --
bench.py reports roofline.traffic only when that tag is the loaded library's.
usage: python tools/make_traffic.py <round dir under gpurun_out or profiles> key=subdir [key=subdir ...]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1]
with open(os.path.join(ROOT, "aquaticgymenv_amd", "lib", "libaqua_hip.so"), "rb") as f:
    tag = hashlib.sha256(f.read()).hexdigest()[:16]
table = {}
for item in sys.argv[2:]:
    key, sub = item.split("=")
    with open(os.path.join(base, sub, "summary.txt")) as f:
        row = json.loads(f.read().strip().splitlines()[-1])
    row["source"] = "profiles/%s/%s/summary.txt (rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; counter units are KB)" % (os.path.basename(base.rstrip("/")), sub)
    row["library_sha16"] = tag
    table[key] = row
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
    json.dump(table, f, indent=1)
print(json.dumps(table, indent=1))
