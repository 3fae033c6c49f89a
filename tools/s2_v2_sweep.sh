for cfg in "2 4 256 16 32 3 2 0" "2 4 128 32 64 3 2 0" "2 4 64 64 128 3 2 0" "2 4 32 128 256 3 2 0" "2 16 512 16 32 3 2 0" "2 16 256 32 64 3 2 0" "2 16 128 64 128 3 2 0" "2 16 64 128 256 3 2 0" "2 16 32 256 512 3 2 0" "3 4 12 128 256 3 2 0"; do
  a=$(python tools/op_bench.py $cfg 0 5 2>/dev/null | grep -i "fwd" | awk '{print $2}'); b=$(URSN_S2CONV_V2=0 python tools/op_bench.py $cfg 0 5 2>/dev/null | grep -i "fwd" | awk '{print $2}'); echo "fwd $cfg v2=$a old=$b"
done
for cfg in "2 4 128 32 16 3 2 1" "2 4 64 64 32 3 2 1" "2 16 256 32 16 3 2 1" "2 16 128 64 32 3 2 1" "2 16 64 128 64 3 2 1" "3 4 96 16 8 3 2 1" "3 4 48 32 16 3 2 1" "3 4 24 64 32 3 2 1"; do
  a=$(python tools/op_bench.py $cfg 0 5 2>/dev/null | grep -i "dgrad" | awk '{print $2}'); b=$(URSN_S2CONV_V2=0 python tools/op_bench.py $cfg 0 5 2>/dev/null | grep -i "dgrad" | awk '{print $2}'); echo "deconv-dgrad $cfg v2=$a old=$b"
done
