timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
