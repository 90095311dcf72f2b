# Quick A/B of library knobs on bench.py (20 and 1024 steps): edit the run lines, then gpurun -- bash tools/bench_sweep.sh
set -e
run() { # label, bench flags, env...
  label=$1; shift; flags=$1; shift
  for steps in "20 5" "1024 32"; do
    env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps ${steps% *} --warmup ${steps#* } --cpu-seconds 0 --no-4k-leg $flags > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_err.txt || { tail -5 gpurun_out/sweep_err.txt; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/sweep_tmp.json').read().strip().splitlines()[-1])
print('$label', 'steps', d['steps'], 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
  done
}
run f3 "--frames-in-flight 3" X=1
run f3pre0 "--frames-in-flight 3" PT_AMD_PREFETCH=0
run f3idle4 "--frames-in-flight 3" PT_AMD_IDLE_MIN=4
run f3idle1 "--frames-in-flight 3" PT_AMD_IDLE_MIN=1
run f3regen16 "--frames-in-flight 3" PT_AMD_REGEN_MIN=16
run f3regen48 "--frames-in-flight 3" PT_AMD_REGEN_MIN=48
run f3ppw64 "--frames-in-flight 3 --pixels-per-wave 64" X=1
run f3ppw256 "--frames-in-flight 3 --pixels-per-wave 256" X=1
