#!/bin/bash
# same-box A/B of whole-library variants: bash tools/lib_ab2.sh tagA tagB ... ("orig" = the built library), 2 rounds
R=$GRAFT_REPO_ROOT
cp $R/u-resnet_amd/csrc/liburesnet_hip.so /tmp/orig.so
for round in 1 2; do for tag in "$@"; do
  if [ "$tag" == "orig" ]; then cp /tmp/orig.so $R/u-resnet_amd/csrc/liburesnet_hip.so; else cp $R/u-resnet_amd/csrc/liburesnet_$tag.so $R/u-resnet_amd/csrc/liburesnet_hip.so; fi
  timeout -k 10 200 python $R/bench.py --steps 8 --warmup 2 --breakdown --no-cpu-baseline > /tmp/ab.log 2>&1
  echo "$tag: $(grep -o 'serialised pass [0-9.]* ms/step), wall [0-9.]*' /tmp/ab.log) | twgradz $(grep '^twgradz' /tmp/ab.log | cut -c25-40) tconv88 $(grep '^tconv<8,8>' /tmp/ab.log | cut -c25-40)"
done; done
cp /tmp/orig.so $R/u-resnet_amd/csrc/liburesnet_hip.so
