# the driver's command, N times on one box: the line's own figures and those of its first five regions
mkdir -p gpurun_out/$1
for i in $(seq 1 ${2:-3}); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$1/driver_$i.json 2> gpurun_out/$1/driver_$i.err || exit 1
  python3 - gpurun_out/$1/driver_$i.json <<'PY'
import json,sys,statistics
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; ev=r["launch_us_regions"]
print(sys.argv[1].split("/")[-1], "regions %d  launch_us %.3f frac %.3f  wall us/step %.3f value %.4g | first five: launch_us %.3f frac %.3f wall %.3f | events by tens:" % (
    d["regions"], r["launch_us"], r["frac"], d["ms_per_step"]*1e3, d["value"], r["first_regions"]["launch_us"], r["first_regions"]["frac"], r["first_regions"]["ms_per_step"]*1e3),
    ["%.2f" % statistics.median(ev[i:i+10]) for i in range(0, len(ev), 10)])
PY
done
