#!/bin/bash
# A/B of library variants u-resnet_amd/csrc/liburesnet_<tag>.so: overlap probe + bench
R=$GRAFT_REPO_ROOT
cp $R/u-resnet_amd/csrc/liburesnet_hip.so /tmp/orig.so
for tag in "$@"; do
  [ "$tag" != "orig" ] && cp $R/u-resnet_amd/csrc/liburesnet_$tag.so $R/u-resnet_amd/csrc/liburesnet_hip.so
  [ "$tag" == "orig" ] && cp /tmp/orig.so $R/u-resnet_amd/csrc/liburesnet_hip.so
  for g in 8192 1024 512; do
    echo -n "$tag ewgrid=$g probe: "; URSN_EW_GRID=$g timeout -k 10 200 python $R/tools/overlap_probe.py 2>&1 | grep alone | sed 's/bn_bwd alone/bn/; s/wgrad alone/wg/'
    URSN_EW_GRID=$g timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --breakdown --no-cpu-baseline > /tmp/ab.log 2>&1
    echo "   bench: $(grep '^bn_bwd' /tmp/ab.log | cut -c25-45) | $(grep '^bn_act' /tmp/ab.log | cut -c25-45) | $(grep -o 'serialised pass [0-9.]* ms/step), wall [0-9.]*' /tmp/ab.log)"
  done
done
cp /tmp/orig.so $R/u-resnet_amd/csrc/liburesnet_hip.so
