# Quick A/B of library knobs on bench.py (20 and 1024 steps): edit the run lines, then gpurun -- bash tools/bench_sweep.sh
set -e
run() { # label, bench flags, env...
  label=$1; shift; flags=$1; shift
  for steps in "20 5" "20 5" "1024 32"; do
    env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps ${steps% *} --warmup ${steps#* } --cpu-seconds 0 --no-4k-leg $flags > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_err.txt || { tail -5 gpurun_out/sweep_err.txt; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/sweep_tmp.json').read().strip().splitlines()[-1])
print('$label', 'steps', d['steps'], 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
  done
}
run base64 "" X=1
run cap128 "" PT_AMD_FRAMES_CAP=128
run cap96 "" PT_AMD_FRAMES_CAP=96
run cap32 "" PT_AMD_FRAMES_CAP=32
run base64 "" X=1
