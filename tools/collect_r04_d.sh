#!/bin/bash
S="--steps 3 --warmup 1"
timeout -k 10 900 bash tools/profile_bench.sh r04_band_100Mx100M $S --n-ref 1e8 --n-unk 1e8 --patches 128 > /dev/null && echo 100M done
du -sh gpurun_out/prof_r04_band_100Mx100M
