#!/bin/bash
# The driver's command twice + the default command once on the box this runs on; one line per run.
#   tools/driver_boxes_r05.sh <tag under gpurun_out> <box label>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/$1; B=$2; mkdir -p $O
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${B}_k20_run$i.json 2>/dev/null; done
python3 bench.py --no-cpu-baseline > $O/${B}_default.json 2>/dev/null
python3 - $O $B <<'PY'
import json, glob, sys
for f in sorted(glob.glob("%s/%s_*.json" % (sys.argv[1], sys.argv[2]))):
    r = json.loads(open(f).read()); ro = r["roofline"]
    print(f.split("/")[-1], "events %.3f us  frac %.4f  live %.4f  wall %.4f  first5 %.4f  outliers %d  slowest x%.2f" % (
        ro["launch_us"], ro["frac"], ro["frac_live"], ro["frac_by_wall"], ro["first_regions"]["frac"],
        r["regions_outliers"]["by_events"]["n"], r["regions_outliers"]["by_events"]["slowest"]["x_median"]))
PY
