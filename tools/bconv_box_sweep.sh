# A/B of the generic bf16 conv kernel's box / buffering choice on the channel-block levels of cfg5 (run on the GPU box)
for shape in "64 32 32" "32 64 64" "16 128 128" "8 256 256"; do
  for box in "" "2,8,32" "1,8,32" "1,4,32" "2,4,32" "4,4,32"; do
    for nb in 2 1; do
      URSN_BCONV_BOX=$box URSN_BCONV_NBUF=$nb python tools/bf16_op_bench.py $shape 2>&1 | grep "^S=" | cut -c1-200
    done
  done
done
