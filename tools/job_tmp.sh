tools/ab_auto2.sh gd "-|" "-|band_grid_div=8" "-|band_grid_div=16" "-|band_grid_div=32"
tools/ab.sh gd "|" "|--set band_grid_div=8" "|--set band_grid_div=16" "|--weights" "|--weights --set band_grid_div=16" "|--n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --steps 5 --warmup 2" "|--n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 --steps 5 --warmup 2 --set band_grid_div=16"
timeout -k 10 600 python -m pytest tests/test_gpu_kernel_parity.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -4
