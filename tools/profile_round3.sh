#!/bin/bash
# Round 3, on the GPU box (gpurun): PMC passes of the path kernel on the configurations SURVEY §8(d).4 asks about --
# glass_stress (config 4), random4096 (config 5), the light-sampling build, and the Cornell box at 4K for comparison --
# through the headless driver (a plain binary: nothing re-execs under rocprofv3), three passes of <= 8 SQ counters each,
# plus one run of the instrumented kernel for the event counters and the path-length histogram of the same samples.
# usage: bash tools/profile_round3.sh        results under gpurun_out/pmc_r3_<config>_<pass>/ ; tools/condense_round3.py makes profiles/r3_pmc_*.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; H=$R/tools/pt_headless
A="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM"
B="SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
C="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
run() {  # name, driver arguments
  name=$1; shift
  rm -rf $R/gpurun_out/pmc_r3_${name}_*
  bash $R/tools/pmc_pass.sh r3_${name}_a "$A" -- $H "$@" || return 1
  bash $R/tools/pmc_pass.sh r3_${name}_b "$B" -- $H "$@" || return 1
  bash $R/tools/pmc_pass.sh r3_${name}_c "$C" -- $H "$@" || return 1
  mkdir -p $R/gpurun_out/pmc_r3_${name}_counters
  timeout -k 10 120 $H "$@" --counters > $R/gpurun_out/pmc_r3_${name}_counters/run.log 2>&1 || { echo "counters run of $name failed"; return 1; }
  echo "$@" > $R/gpurun_out/pmc_r3_${name}_counters/args.txt
  tail -2 $R/gpurun_out/pmc_r3_${name}_a/run.log | cut -c1-200
}
run glass --scene glass_stress --width 1280 --height 720 --spp 32 --spp-per-launch 16 --max-depth 32 &&
run random4096 --scene random4096 --width 3840 --height 2160 --spp 2 --spp-per-launch 1 --max-depth 16 &&
run nee --scene cuda_default --width 1280 --height 720 --spp 32 --spp-per-launch 16 --mis &&
run cornell720 --scene cuda_default --width 1280 --height 720 --spp 32 --spp-per-launch 16 &&
run cornell4k --scene cuda_default --width 3840 --height 2160 --spp 32 --spp-per-launch 16 &&
echo "profile_round3 done"
