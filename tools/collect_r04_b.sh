#!/bin/bash
# Round 4, call B: rocprofv3 stats + FETCH/WRITE + SQ passes of the 10M-class configurations.
S="--steps 5 --warmup 2"
timeout -k 10 300 bash tools/profile_bench.sh r04_band_10Mx10M $S > /dev/null && echo headline done
timeout -k 10 300 bash tools/profile_bench.sh r04_band_10Mx10M_weighted $S --weights > /dev/null && echo weighted done
timeout -k 10 300 bash tools/profile_bench.sh r04_band_10Mx10M_kpc $S --kpc > /dev/null && echo kpc done
timeout -k 10 300 bash tools/profile_bench.sh r04_band_10Mx10M_rweight $S --rweight -1 > /dev/null && echo rweight done
timeout -k 10 300 bash tools/profile_bench.sh r04_band_1Mx1M --steps 20 --warmup 5 --n-ref 1e6 --n-unk 1e6 --patches 16 > /dev/null && echo 1M done
du -sh gpurun_out/prof_r04_*
