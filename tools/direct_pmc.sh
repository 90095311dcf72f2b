set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/direct_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/valu -- python3 $R/tools/direct_run.py > $O/valu.log 2>&1 || { echo fail1; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 GRBM_GUI_ACTIVE --output-format csv -d $O/wait -- python3 $R/tools/direct_run.py > $O/wait.log 2>&1 || { echo fail2; tail -5 $O/wait.log; exit 1; }
echo done
