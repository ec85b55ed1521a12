#!/bin/bash
# kernel timeline of a region with the done-mask exchange: where do the steps lose their time?
R=$PWD
O=$R/gpurun_out/r03_trace_exchange
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --steps 1000 --warmup 100 --regions 2 --force-exchange --exchange ipc > $O/bench.json 2> $O/err.log
cd $R
python3 - <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r03_trace_exchange/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
steps = [e for e in ev if "step_ns_kernel" in e[2]]
copies = [e for e in ev if "copy_words" in e[2]]
print("step kernels %d, copy kernels %d" % (len(steps), len(copies)))
import statistics
d = [e[1] - e[0] for e in steps]
print("step duration ns: median %d mean %.0f" % (statistics.median(d), statistics.mean(d)))
gaps = [steps[i + 1][0] - steps[i][1] for i in range(len(steps) - 1)]
print("gap between consecutive steps ns: median %d mean %.0f; gaps > 20 us: %d" % (statistics.median(gaps), statistics.mean(gaps), sum(g > 20000 for g in gaps)))
for c in copies:
    # steps overlapping / right around this copy
    near = [(s[0] - c[0], s[1] - s[0]) for s in steps if abs(s[0] - c[0]) < 300000]
    print("copy: start %d dur %.1f us queue %s; steps within +-300 us: n=%d, mean dur %.0f ns" % (c[0] - ev[0][0], (c[1] - c[0]) / 1e3, c[3], len(near), statistics.mean(x[1] for x in near) if near else 0))
# windows: mean step period in 100-step windows
per = []
for i in range(0, len(steps) - 100, 100):
    per.append((steps[i + 100][0] - steps[i][0]) / 100.0)
print("step period per 100-step window (ns):", " ".join("%.0f" % p for p in per))
PY
