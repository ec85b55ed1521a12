#!/bin/bash
# shader/memory clocks and power while the benchmark runs (read-only)
R=${GRAFT_REPO_ROOT:-$PWD}
python3 $R/bench.py --steps 6000000 --warmup 200 --no-cpu-baseline > $R/gpurun_out/clk_bench.json 2> $R/gpurun_out/clk_bench.err &
BP=$!
sleep 20
for i in 1 2 3; do
  rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
  for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f 2>/dev/null | head -5; done
  sleep 2
done
wait $BP
cat $R/gpurun_out/clk_bench.json
