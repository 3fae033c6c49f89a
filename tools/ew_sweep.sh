#!/bin/bash
# A/B of elementwise-kernel variants: swaps the library (built with -DURSN_EW_NT=n) and grid caps
R=$GRAFT_REPO_ROOT
cp $R/u-resnet_amd/csrc/liburesnet_hip.so /tmp/orig.so
for nt in 0 1 2 3; do
  cp $R/u-resnet_amd/csrc/liburesnet_nt$nt.so $R/u-resnet_amd/csrc/liburesnet_hip.so
  for g in 2048 4096 8192 16384; do
    URSN_EW_GRID=$g timeout -k 10 200 python $R/bench.py --steps 3 --warmup 1 --breakdown --no-cpu-baseline > /tmp/ew.log 2>&1
    echo "nt=$nt grid=$g $(grep '^bn_bwd' /tmp/ew.log | cut -c1-50) | $(grep '^bn_act' /tmp/ew.log | cut -c25-50) | $(grep -o 'wall [0-9.]*' /tmp/ew.log)"
  done
done
cp /tmp/orig.so $R/u-resnet_amd/csrc/liburesnet_hip.so
