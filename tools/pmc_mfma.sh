#!/bin/bash
# MFMA-busy / issue / wait counters per kernel for the default bench workload (cfg3, weight-gradient stream serialised so
# that kernel intervals do not overlap).  Counters only with --kernel-trace (no other trace domain).  On the GPU box:
#   bash tools/pmc_mfma.sh && python tools/pmc_mfma.py > profiles/r02_pmc_mfma_busy_cfg3.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3_3d192_f8_b4}
URSN_WGRAD_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -o m -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_mfma.log 2>&1
