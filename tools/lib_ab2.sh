#!/bin/bash
# A/B of two builds of the library on ONE box: tools/lib_ab2.sh <cmd...>  runs <cmd> with liburesnet_old.so then the current one, twice.
R=$GRAFT_REPO_ROOT
C=$R/u-resnet_amd/csrc
cp $C/liburesnet_hip.so /tmp/new.so
for rep in 1 2; do
  cp $C/liburesnet_old.so $C/liburesnet_hip.so; echo "== old"; "$@"
  cp /tmp/new.so $C/liburesnet_hip.so; echo "== new"; "$@"
done
