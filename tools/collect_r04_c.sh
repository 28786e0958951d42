#!/bin/bash
# Round 4, call C: the passes of the large configurations (config #5, 100M x 100M, config #4).
S="--steps 3 --warmup 1"
timeout -k 10 500 bash tools/profile_bench.sh r04_band_50Mx50M_3scales $S --n-ref 5e7 --n-unk 5e7 --patches 128 --scales 3 > /dev/null && echo config5 done
timeout -k 10 400 bash tools/profile_bench.sh r04_autocorr_10M_100M $S --n-ref 1e7 --auto-randoms 1e8 --weights > /dev/null && echo config4 done
du -sh gpurun_out/prof_r04_*
