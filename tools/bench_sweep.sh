# Quick A/B of library knobs on bench.py (20 and 1024 steps): edit the run lines, then gpurun -- bash tools/bench_sweep.sh
set -e
run() { # label, bench flags, env...
  label=$1; shift; flags=$1; shift
  for steps in "20 5" "1024 32" "1024 32"; do
    env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps ${steps% *} --warmup ${steps#* } --cpu-seconds 0 --no-4k-leg --no-direct-leg $flags > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_err.txt || { tail -5 gpurun_out/sweep_err.txt; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/sweep_tmp.json').read().strip().splitlines()[-1])
print('$label', 'steps', d['steps'], 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
  done
}
run base_f3 "" X=1
run f2 "--frames-in-flight 2" X=1
run f3 "--frames-in-flight 3" X=1
run f2 "--frames-in-flight 2" X=1
run f4 "--frames-in-flight 4" X=1
run f2_4k "--frames-in-flight 2 --workload cornell_4k_16spp_per_step --steps 64" X=1
run f3_4k "--frames-in-flight 3 --workload cornell_4k_16spp_per_step --steps 64" X=1
