cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
O=$PWD/gpurun_out/r3g
( cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/new -- python3 $GRAFT_REPO_ROOT/tools/enc_time.py rle8_packed_multi 0 8 > $O/new.log 2>&1
  HSRLE_LIB=$GRAFT_REPO_ROOT/variants/libhsrle_r02.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/old -- python3 $GRAFT_REPO_ROOT/tools/enc_time.py rle8_packed_multi 0 8 > $O/old.log 2>&1
  cd $GRAFT_REPO_ROOT
  for v in new old; do echo "== $v"; cat $O/$v/*/*_kernel_stats.csv | cut -c1-200 | head -12; done
  bash tools/pmc_kernel.sh r3g k_encode8_blocks -- tools/enc_time.py rle8_packed_multi 0 8
) > gpurun_out/r3g/log.txt 2>&1
cat gpurun_out/r3g/log.txt
