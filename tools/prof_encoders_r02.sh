#!/bin/bash
# usage (GPU box, via gpurun): bash tools/prof_encoders_r02.sh -> gpurun_out/enc_r02/summary.txt: rocprofv3 kernel rows of the round-2 encoders (2 GiB each)
export TMPDIR=/tmp
O=gpurun_out/enc_r02; mkdir -p $O; : > $O/summary.txt
for spec in "rle8_single 0" "rle8_single 1" "rle8_single_short 0" "rle128_sym 0" "rle128_byte_packed 1" "rle8_packed_multi 1" "rle8_packed_multi 0" "rle64_3symlut_byte 1" "rle16_sym 1"; do
  set -- $spec
  d=$O/$1_$2
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/enc_time.py $1 $2 2 > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $1 kind $2 (2 GiB, 4 KiB blocks): $(grep -o 'encode ms.*' $d.log | tail -1)" >> $O/summary.txt
  python3 tools/show_stats.py $f | grep -v "k_synth\|elementwise\|fillBuffer\|copyBuffer" | head -8 >> $O/summary.txt
done
cat $O/summary.txt
