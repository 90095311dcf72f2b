#!/bin/bash
# Runs on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and PMC passes for the bench workload.
# usage: bash tools/profile_round.sh <tag>      results under gpurun_out/<tag>/
set -u
tag=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/bench.py
timeout -k 10 300 python3 -u $B > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
timeout -k 10 300 python3 -u $B --frames-in-flight 1 --cpu-seconds 0 > $O/bench_serial.json 2> $O/bench_serial.err || { echo "serial bench failed"; exit 1; }
timeout -k 10 300 python3 -u $B --workload cornell_4k_16spp_per_step --steps 24 --warmup 4 --cpu-seconds 0 > $O/bench_4k.json 2> $O/bench_4k.err || { echo "4k bench failed"; exit 1; }
# kernel trace + stats of the same bench command (default = frames in flight 8, and strictly serial)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $B --cpu-seconds 0 > $O/trace_default.log 2>&1 || { echo "trace default failed"; tail -5 $O/trace_default.log; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $B --frames-in-flight 1 --cpu-seconds 0 > $O/trace_serial.log 2>&1 || { echo "trace serial failed"; exit 1; }
# PMC passes (each its own run; <= 8 SQ / 4 TCC counters per pass), serial launches so that per-dispatch numbers are per launch
P="python3 $B --frames-in-flight 1 --steps 64 --warmup 8 --cpu-seconds 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_valu -- $P > $O/pmc_valu.log 2>&1 || { echo "pmc valu failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_wait -- $P > $O/pmc_wait.log 2>&1 || { echo "pmc wait failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
echo "profile_round done"
