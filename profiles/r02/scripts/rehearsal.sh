#!/bin/bash
# what the driver does at round end, on one box: GPU tests, smoke(), the bench with its flags
set -e
mkdir -p gpurun_out/r02z
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02z/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r02z/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/r02z/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z/bench_driver.json 2> gpurun_out/r02z/bench_driver.err || { tail gpurun_out/r02z/bench_driver.err; exit 1; }
python - <<'PY'
import json
r=json.load(open('gpurun_out/r02z/bench_driver.json'))
print("value %.4g ms/step %.6f launch_us %.3f frac %.4f traffic %s cpu_baseline %s" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"], r["roofline"]["frac"], r["roofline"]["traffic"], r["cpu_baseline"]["value"]))
PY
