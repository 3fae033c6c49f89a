#!/bin/bash
# What bounds tdeconv_kernel<16,8>?  URSN_TD_ABLATE bits: 1 no output stores, 2 no MFMAs, 4 no global loads of the staged planes,
# 8 no LDS exchange reads (measurement-only instrumentation, see tools/patches/README.md; results are wrong, only the time is meaningful).
for ab in 0 1 2 3 4 6 7 8 10; do
  t=$(URSN_TD_ABLATE=$ab python tools/op_bench.py 3 4 96 16 8 3 2 1 0 5 2>/dev/null | grep -i "fwd" | awk '{print $2}')
  echo "ablate=$ab fwd=$t ms"
done
