cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ah
( export HSRLE_LIB=$PWD/variants/libhsrle_exp.so
  for key in rle8_packed_multi rle64_3symlut_byte rle32_sym rle16_7symlut_byte rle8_3symlut_short; do for rl in 1 2; do HSRLE_RUNLIST=$rl timeout 600 python tools/scratch/threshold.py $key 2>&1 | grep -v amdgpu.ids; done; done
) > gpurun_out/r3ah/log.txt 2>&1
python3 - <<'PY'
import collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/r3ah/log.txt'):
    f=l.split()
    if len(f)>8 and f[2]=='MiB':
        rows[(f[1],int(f[3]),int(f[5]))][f[0]]=float(f[8])
for k in sorted(rows): print(k, rows[k], 'runlist/other = %.2f'%(rows[k].get('1',0)/max(rows[k].get('2',1),1e-9)))
PY
