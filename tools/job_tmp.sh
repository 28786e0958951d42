tools/ab_auto2.sh w16 "-|" "w16|" "-|" "w16|"
C5="--n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --steps 5 --warmup 2"
tools/ab.sh c5fix "|$C5" "prehalf|$C5"
