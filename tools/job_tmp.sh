timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
tools/ab.sh rw "|--rweight -1" "|--rweight -1 --weights" "|" "|--rweight -1 --kpc"
