#!/bin/bash
# the store policy of large batches (STORE_WB_*): the shipped library against the round's first build ('old' variant), by size
set -e
python tools/ab.py --rounds 3 old@2 default@2 old@1 default@1 old@0 default@0
for n in 1048576 2097152 4194304 8388608 12582912 16777216; do
    echo "== $n worlds"
    python tools/ab.py --rounds 2 --envs $n --steps 200 old@2 default@2 old@1 default@1
done
