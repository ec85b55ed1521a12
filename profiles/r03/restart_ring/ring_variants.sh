#!/bin/bash
O=gpurun_out/r03_ring_variants.txt
python3 tools/r03/ring_check.py > $O 2>&1 || { tail -5 $O; exit 1; }
for v in "$@"; do
  echo "== $v" >> $O
  AQUA_HIP_LIB=$PWD/aquaticgymenv_amd/lib/variants/libaqua_hip_$v.so python3 tools/r03/ring_check.py 2>&1 | grep "ring\|Error\|rror" >> $O
done
cat $O
