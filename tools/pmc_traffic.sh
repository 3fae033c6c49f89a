#!/bin/bash
# HBM traffic per kernel for the default bench workload: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md, HBM section).  Run on the GPU box:  bash tools/pmc_traffic.sh ; python tools/pmc_traffic.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3_3d192_f8_b4}
rm -rf $R/gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  URSN_WGRAD_STREAM=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic/$c -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_traffic_$c.log 2>&1 || exit 1
done
