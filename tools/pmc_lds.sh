#!/bin/bash
# LDS bank-conflict cycles per kernel (rocprofv3 --pmc with --kernel-trace only).  bash tools/pmc_lds.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3_3d192_f8_b4}
URSN_WGRAD_STREAM=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds -o l -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_lds.log 2>&1
