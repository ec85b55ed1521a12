mkdir -p gpurun_out/r04/ramp
for i in 1 2; do
for r in 1 10 30 100 300; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --ramp-replays $r --cpu-seconds 1 > gpurun_out/r04/ramp/ramp${r}_$i.json 2> gpurun_out/r04/ramp/ramp${r}_$i.err || exit 1
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04/ramp/ramp${r}_$i.json').read().strip().splitlines()[-1])
print('ramp $r', 'launch_us %.3f frac %.3f wall %.3f' % (d['roofline']['launch_us'], d['roofline']['frac'], d['ms_per_step']*1e3), [round(v,2) for v in d['roofline']['launch_us_regions']], [round(v*1e3/20,2) for v in d['regions_ms']])
"
done
done
