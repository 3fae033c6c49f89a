#!/bin/bash
# kernel-trace of one bf16 conv layer shape: tools/trace_op.sh TAG <bf16_op_bench args...>; prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$TAG -- python3 $R/tools/bf16_op_bench.py "$@" > $R/gpurun_out/trace_$TAG.log 2>&1 || exit 1
grep "S=" $R/gpurun_out/trace_$TAG.log
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/trace_$TAG/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
