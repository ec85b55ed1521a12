#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python examples/bearing_policy.py > $O/example_bearing.log 2>&1 || { tail -20 $O/example_bearing.log; exit 1; }
tail -2 $O/example_bearing.log
python examples/dqn_replay.py > $O/example_dqn.log 2>&1 || { tail -20 $O/example_dqn.log; exit 1; }
tail -2 $O/example_dqn.log
python tools/soak.py > $O/soak.log 2>&1 || { tail -20 $O/soak.log; exit 1; }
tail -2 $O/soak.log
