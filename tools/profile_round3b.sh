#!/bin/bash
# Round 3 refresh of tools/profile_round2.sh (same passes; the FP64 direct leg is left out of the traced and counted commands).
# Runs on the GPU box (gpurun): the bench lines, rocprofv3 kernel traces of the same commands, PMC passes.
# usage: bash tools/profile_round3b.sh <tag>      results under gpurun_out/<tag>/ ; condensed by the caller into profiles/
set -u
tag=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/bench.py
DRV="--gpus 1 --steps 20 --warmup 5"      # the driver's arguments
# 1. the driver's command as it is (both legs, cpu baseline), and the default command
timeout -k 10 300 python3 -u $B $DRV > $O/bench_driver_args.json 2> $O/bench_driver_args.err || { echo "bench (driver args) failed"; tail -5 $O/bench_driver_args.err; exit 1; }
timeout -k 10 300 python3 -u $B --cpu-seconds 0 > $O/bench_default.json 2> $O/bench_default.err || { echo "bench (default) failed"; exit 1; }
# 2. kernel traces: the headline leg with the driver's arguments, the 4K leg, the default (1024 steps)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_driver_args -- python3 $B $DRV --cpu-seconds 0 --no-4k-leg --no-direct-leg > $O/trace_driver_args.log 2>&1 || { echo "trace (driver args) failed"; tail -5 $O/trace_driver_args.log; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_4k -- python3 $B --workload cornell_4k_16spp_per_step --steps 64 --warmup 2 --cpu-seconds 0 --no-4k-leg --no-direct-leg > $O/trace_4k.log 2>&1 || { echo "trace (4k) failed"; exit 1; }
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $B --cpu-seconds 0 --no-4k-leg --no-direct-leg > $O/trace_default.log 2>&1 || { echo "trace (default) failed"; exit 1; }
# (one launch per frame, strictly serial: render-ahead off for this command and for the serial PMC passes below; the same
#  loop as the library runs it by default -- the next frame rendered ahead -- is traced after them)
export PT_AMD_RENDER_AHEAD=0
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -- python3 $B --frames-in-flight 1 --steps 128 --warmup 8 --cpu-seconds 0 --no-4k-leg --no-direct-leg > $O/trace_serial.log 2>&1 || { echo "trace (serial) failed"; exit 1; }
# 3. PMC passes (each its own run; <= 8 SQ / 4 TCC counters per pass), serial launches so that per-dispatch numbers are per launch
P="python3 $B --frames-in-flight 1 --steps 64 --warmup 8 --cpu-seconds 0 --no-4k-leg --no-direct-leg"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_valu -- $P > $O/pmc_valu.log 2>&1 || { echo "pmc valu failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_wait -- $P > $O/pmc_wait.log 2>&1 || { echo "pmc wait failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > $O/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > $O/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
unset PT_AMD_RENDER_AHEAD
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ahead -- python3 $B --frames-in-flight 1 --steps 128 --warmup 8 --cpu-seconds 0 --no-4k-leg --no-direct-leg > $O/trace_ahead.log 2>&1 || { echo "trace (render-ahead) failed"; exit 1; }
# 4. the same four passes on the driver's command: its 20 timed frames are ONE launch of the path kernel (frames in flight collect
#    them) followed by one fold pass; rocprofv3 serialises dispatches while it counts, which changes nothing for a single launch
P2="python3 $B $DRV --cpu-seconds 0 --no-4k-leg --no-direct-leg"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmcf_valu -- $P2 > $O/pmcf_valu.log 2>&1 || { echo "pmc valu (frames) failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_fetch -- $P2 > $O/pmcf_fetch.log 2>&1 || { echo "pmc fetch (frames) failed"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcf_write -- $P2 > $O/pmcf_write.log 2>&1 || { echo "pmc write (frames) failed"; exit 1; }
echo "profile_round3b done"
