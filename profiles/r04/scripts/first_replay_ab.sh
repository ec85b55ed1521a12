# region 0 of the driver's command: hipGraphUpload alone against one untimed replay that is rolled back
mkdir -p gpurun_out/r04/first_replay
for i in 1 2 3 4; do
  for mode in upload rollback; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --first-replay $mode > gpurun_out/r04/first_replay/${mode}_$i.json 2> gpurun_out/r04/first_replay/${mode}_$i.err || exit 1
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/first_replay/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "launch_us %.3f frac %.3f wall us/step %.3f" % (d["roofline"]["launch_us"], d["roofline"]["frac"], d["ms_per_step"]*1e3),
          "regions_ms", ["%.4f" % v for v in d["regions_ms"]], "event regions", ["%.2f" % v for v in d["roofline"]["launch_us_regions"]], d["config"]["timed_graph_first_replay"][:40])
PY
