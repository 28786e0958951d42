#!/bin/bash
# Round 4, call A on the GPU box: the GPU test log and the bench lines of every supported configuration (one box).
O=gpurun_out/round_r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/r04_gpu_tests.log 2>&1; tail -3 $O/r04_gpu_tests.log
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -1 $O/r04_smoke.log
bash tools/collect_round.sh r04
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --n-ref 1e7 --auto-randoms 1e8 --weights > $O/r04_autocorr_10M_100M_weighted_bench.json 2> $O/autocorr.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/round_r04/r04_autocorr_10M_100M_weighted_bench.json").read().strip().splitlines()[-1])
print("autocorr ms/step", round(d["ms_per_step"],3), {k: round(v["count_kernel_ms"],3) for k,v in d["roofline"]["counts"].items()}, "fixed", round(d["roofline"]["fixed_cost_ms"],3))
PY
python tools/probe_api_overhead.py 1e6 2>&1 | grep -E "warm:" > $O/r04_api_overhead.txt; cat $O/r04_api_overhead.txt
# three ranks on the one GPU, gloo: the multi-GPU line with its self-verifying record (no --scaling probe: 50M x 50M x 3 ranks does not fit the minute)
YAW_BENCH_BACKEND=gloo YAW_AMD_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 3 --steps 10 --warmup 3 --no-probe > $O/r04_bench_gpus3_gloo_one_gpu_rehearsal.json 2> $O/rehearsal.err; tail -c 1500 $O/r04_bench_gpus3_gloo_one_gpu_rehearsal.json
