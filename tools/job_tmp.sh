timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
tools/ab_auto2.sh dq "-|"
tools/ab.sh dq "|" "|--weights" "|--kpc" "|--n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --steps 5 --warmup 2" "|--n-ref 1e8 --n-unk 1e8 --patches 128 --steps 5 --warmup 2" "|--n-ref 1e6 --n-unk 1e6 --patches 16 --steps 50 --warmup 10"
