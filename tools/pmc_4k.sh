#!/bin/bash
# PMC passes at 3840x2160, 16 spp per launch (2 launches) through the headless driver; on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; H=$R/tools/pt_headless; A="--width 3840 --height 2160 --spp 32 --spp-per-launch 16"
rm -rf $R/gpurun_out/pmc_4k_*; 
bash $R/tools/pmc_pass.sh 4k_a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" -- $H $A &&
bash $R/tools/pmc_pass.sh 4k_b "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" -- $H $A &&
bash $R/tools/pmc_pass.sh 4k_c "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" -- $H $A &&
for p in a b c; do python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_4k_$p > $R/gpurun_out/pmc_4k_$p.json; done; tail -1 $R/gpurun_out/pmc_4k_a/run.log
