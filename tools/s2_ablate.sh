#!/bin/bash
# What bounds s2conv_kernel?  Times the forward of the stride-2 conv with parts of the kernel switched off (URSN_S2_ABLATE bits:
# 1 no MFMAs, 2 no LDS stores of the staged halo, 4 no global loads of it; results are wrong, only the time is meaningful).
for cfg in "3 4 192 8 16 3 2 0" "3 4 96 16 32 3 2 0" "3 4 48 32 64 3 2 0" "2 16 256 32 64 3 2 0"; do
  for ab in 0 1 2 4 6 7; do
    t=$(URSN_S2CONV_V2=0 URSN_S2_ABLATE=$ab python tools/op_bench.py $cfg 0 5 2>/dev/null | grep -i "fwd" | awk '{print $2}')
    echo "$cfg ablate=$ab fwd=$t ms"
  done
done
