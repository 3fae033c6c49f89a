for kb in 78 106 130 158; do
  for shape in "64 32 32" "32 64 64" "16 128 128" "64 64 32"; do
    URSN_BCONV_LDS_KB=$kb python tools/bf16_op_bench.py $shape 2>&1 | grep "^S="
  done
done
