#!/bin/bash
# GPU suite + the driver's bench command; results under gpurun_out/<tag>/
set -u
tag=${1:-r3chk}; O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gputests.log 2>&1; rc=$?
tail -8 $O/gputests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err || { tail -5 $O/bench_driver_args.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_args.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("driver args:", d["value"], "Msamples/s", d["ms_per_step"], "ms/step frac", r["frac"], "serial ms", r.get("kernel_ms_serial"), "frac_serial", r.get("frac_serial"))
l=d.get("legs",{}).get("cornell_4k_1024spp_strong_split"); print("4k leg:", l and (l["value"], l["roofline"]["frac"])); print("cpu:", d.get("cpu_baseline",{}).get("value"))
PY
