#!/bin/bash
# A/B of library builds on bench.py itself, interleaved:  bash tools/bench_ab.sh "<bench flags>" libA libB ...   (main = the tree's library)
set -u
flags=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset PT_AMD_LIB; else export PT_AMD_LIB=$PWD/path_tracer_amd/lib/libpt_amd_$v.so; fi
  timeout -k 10 200 python bench.py --gpus 1 --cpu-seconds 0 --no-4k-leg --no-direct-leg $flags > gpurun_out/ab_tmp.json 2> gpurun_out/ab_err.txt || { tail -5 gpurun_out/ab_err.txt; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$v', '[$flags]', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'serial_ms', r.get('kernel_ms_serial'), 'frac_serial', r.get('frac_serial'))"
done
done
