# A/B of library knobs on bench.py: the Cornell box (20 and 1 024 steps), 4K, and the 150- and 1 000-primitive scenes the
# launch heuristics were fitted on (pt_api.cpp, namespace tune), so that a kernel change that moves an optimum shows as a
# number.  Edit the run lines, then:  gpurun -- bash tools/bench_sweep.sh
set -e
run() { # label, bench flags, env...
  label=$1; shift; flags=$1; shift
  for steps in "20 5" "1024 32" "1024 32"; do
    env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps ${steps% *} --warmup ${steps#* } --cpu-seconds 0 --no-4k-leg --no-direct-leg $flags > gpurun_out/sweep_tmp.json 2>gpurun_out/sweep_err.txt || { tail -5 gpurun_out/sweep_err.txt; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/sweep_tmp.json').read().strip().splitlines()[-1])
print('$label', 'steps', d['steps'], 'value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'serial_ms', d['roofline'].get('kernel_ms_serial'))"
  done
}
run f2 "--frames-in-flight 2" X=1
run f3 "--frames-in-flight 3" X=1
run f2_4k "--frames-in-flight 2 --workload cornell_4k_16spp_per_step --steps 64" X=1
run f2_150 "--frames-in-flight 2 --workload spheres150_720p_1spp_per_step" X=1
run f2_150_regen48 "--frames-in-flight 2 --workload spheres150_720p_1spp_per_step" PT_AMD_REGEN_MIN=48
run f2_1000 "--frames-in-flight 2 --workload spheres1000_360p_1spp_per_step" X=1
run f2_1000_regen48 "--frames-in-flight 2 --workload spheres1000_360p_1spp_per_step" PT_AMD_REGEN_MIN=48
run f2_1000_grid75 "--frames-in-flight 2 --workload spheres1000_360p_1spp_per_step" PT_AMD_GRID_PCT=75
