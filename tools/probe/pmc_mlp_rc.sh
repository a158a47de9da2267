#!/bin/bash
# PMC stall breakdown of the mlp_rc kernels (run on the GPU box):  bash tools/probe/pmc_mlp_rc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mlp_rc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/p1 -o p1 --output-format csv -- python3 $R/tools/probe/mlp_rc_pmc_run.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $O/p2 -o p2 --output-format csv -- python3 $R/tools/probe/mlp_rc_pmc_run.py > $O/p2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/tools/probe/mlp_rc_pmc_run.py > $O/kt.log 2>&1
python3 $R/tools/probe/pmc_stalls.py $O
grep -h "mlp_rc\|rc_reduce\|chan_reduce" $O/kt/*kernel_stats.csv 2>/dev/null || find $O/kt -name "*stats*" | head
