#!/bin/bash
# The other workloads of bench.py (BASELINE configs 4 and 5, the light-sampling build) -> gpurun_out/<tag>/bench_*.json
set -u
tag=${1:-r3w}; O=gpurun_out/$tag; mkdir -p $O
run() { name=$1; shift; timeout -k 10 300 python bench.py --gpus 1 --cpu-seconds 0 --no-4k-leg --no-direct-leg "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name', d['value'], 'Msamples/s', d['mrays_per_s'], 'Mrays/s frac', r['frac'], 'serial frac', r.get('frac_serial'))"; }
run glass --workload glass_stress_720p_16spp_per_step --steps 64 --warmup 4
run random4096 --workload random4096_4k_1spp_per_step --steps 8 --warmup 1
run mis --mis
run mis_driver_args --mis --steps 20 --warmup 5
