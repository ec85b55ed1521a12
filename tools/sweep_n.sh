#!/bin/bash
# per-step time against batch size: whole kernel, stepping blocks alone (nw), re-seeding blocks alone (nm)
for n in 65536 131072 196608 262144 327680 393216 524288; do echo "N=$n"; timeout -k 10 200 python tools/ab.py --rounds 1 --envs $n default@2 nw@2 nm@2 2>&1 | grep us/step; done
