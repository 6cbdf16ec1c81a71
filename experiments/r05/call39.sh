#!/bin/bash
# round 5 call 39: encode throughput of Short codecs on the position-parallel encoder, 4 GiB (r04 sweep: 1204 / 1289 / 1440 / 1611 runs, 1139 / 1138 / 1254 / 968 video)
cd /root/repo
python tools/mini_sweep.py 4096 rle16_sym_short,rle24_1symlut_sym_short,rle32_1symlut_byte_short,rle64_byte_short,rle32_sym_packed 2>&1 | grep -v random
