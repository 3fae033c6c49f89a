#!/bin/bash
# Round profiles (run on the GPU box from the repo root): rocprofv3 kernel statistics of the bench command for cfg3 and cfg5 with
# the weight-gradient stream serialised (kernel intervals do not overlap), and a roctx marker trace of cfg3 attributed to layers.
#   bash tools/profile_round.sh r03      -> gpurun_out/prof_<round>/..., summaries copied to profiles/ by the caller
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03}
O=$R/gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
URSN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3 -o k -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/cfg3.log 2>&1 || exit 1
URSN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -o k -- python3 $R/bench.py --workload cfg5_3d256_f8_b4_bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg5.log 2>&1 || exit 1
URSN_ROCTX=1 URSN_WGRAD_STREAM=0 rocprofv3 --kernel-trace --hip-trace --marker-trace --output-format csv -d $O/roctx -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/roctx.log 2>&1 || exit 1
cp $(ls $O/cfg3/*kernel_stats.csv $O/cfg3/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${T}_rocprofv3_kernel_stats_cfg3_serial.csv
cp $(ls $O/cfg5/*kernel_stats.csv $O/cfg5/*/*kernel_stats.csv 2>/dev/null | head -1) $O/${T}_rocprofv3_kernel_stats_cfg5_bf16_serial.csv
python3 $R/tools/marker_attrib.py $O/roctx > $O/${T}_roctx_attribution_cfg3.csv || exit 1
sed -n 1,5p $O/${T}_roctx_attribution_cfg3.csv
# keep the merge-back small: the raw traces stay on the box
rm -rf $O/cfg3 $O/cfg5 $O/roctx
