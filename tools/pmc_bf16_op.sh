#!/bin/bash
# PMC passes for one bf16 conv layer shape: tools/pmc_bf16_op.sh TAG <op_bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
  g=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$g -- python3 $R/tools/bf16_op_bench.py "$@" > $R/gpurun_out/pmc_${TAG}_$g.log 2>&1 || exit 1
done
