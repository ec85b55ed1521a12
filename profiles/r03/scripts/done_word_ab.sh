#!/bin/bash
# the done-mask word written through (agent scope) up to 2 M worlds: new default against the previous build ('old' variant)
set -e
python tools/ab.py --rounds 3 old@2 default@2 old@1 default@1 old@0 default@0
python tools/ab.py --rounds 2 --envs 1048576 --steps 1000 old@2 default@2
python tools/ab.py --rounds 2 --envs 2097152 --steps 500 old@2 default@2
python tools/ab.py --rounds 2 --envs 4194304 --steps 400 old@2 default@2
python tools/ab.py --rounds 2 --envs 16777216 --steps 200 old@2 default@2
python tools/ab.py --rounds 2 --tables old@2 default@2 old@1 default@1
