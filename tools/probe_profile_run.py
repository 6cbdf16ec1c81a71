# decode/encode only, for profiling: python tools/probe_profile_run.py [size_mib] [bs] [codec] [iters]
import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=(int(sys.argv[1]) if len(sys.argv)>1 else 1024)<<20
bs=int(sys.argv[2]) if len(sys.argv)>2 else 4096
codec=sys.argv[3] if len(sys.argv)>3 else 'rle8_packed_multi'
iters=int(sys.argv[4]) if len(sys.argv)>4 else 3
S={'8':1,'16':2,'24':3,'32':4,'48':6,'64':8,'128':16}[codec.split('_')[0][3:]]
src=hsrle.synth(0,S,2,size); torch.cuda.synchronize()
dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda')
ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(1,dtype=torch.int32,device='cuda')
for _ in range(iters):
    hsrle.compress_async(codec,src,dst,bs,workspace=ws)
torch.cuda.synchronize()
info=hsrle.container_info(dst)
for _ in range(iters):
    hsrle.decompress_async(dst,info,out,st)
torch.cuda.synchronize()
print('ratio',info.totalSize/size,'ok',int(st.item())==0 and torch.equal(out,src))
