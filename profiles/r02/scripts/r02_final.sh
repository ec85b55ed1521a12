#!/bin/bash
# what the driver runs at round end, on one box: the GPU suite, smoke(), the bench with its flags
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02final
mkdir -p $O
cd $R
sha256sum aquaticgymenv_amd/lib/libaqua_hip.so | cut -c1-16
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || { tail $O/bench_driver.err; exit 1; }
cat $O/bench_driver.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
python3 - $O/bench_default.json <<'PY'
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default: value %.4g ms/step %.6f launch_us %.3f frac %.4f traffic %s host %s" % (r["value"], r["ms_per_step"], r["roofline"]["launch_us"], r["roofline"]["frac"], r["roofline"]["traffic"], r.get("host")))
PY
