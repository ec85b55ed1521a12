for n in 98304 196608 229376 262144 294912 393216 524288; do echo "N=$n"; timeout -k 10 120 python tools/ab.py --rounds 1 --envs $n default@2 default@0 2>&1 | grep us/step; done
