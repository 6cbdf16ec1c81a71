import sys, os, random
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import *
rng=random.Random(1)
d=mixed_runs(rng, 30000)+bytes(rng.randrange(256) for _ in range(5000))+mixed_runs(rng,5000)
key=sys.argv[1] if len(sys.argv)>1 else 'rle8_packed_multi'; bs=int(sys.argv[2]) if len(sys.argv)>2 else 512
src=torch.frombuffer(bytearray(d),dtype=torch.uint8).cuda()
cont,info=hsrle.compress(key,src,block_size=bs)
ci,streams=hsrle.split_container(cont.cpu().numpy().tobytes())
out=torch.full((len(d),),0xEE,dtype=torch.uint8,device='cuda'); st=torch.zeros(64,dtype=torch.int32,device='cuda')
hsrle.decompress_async(cont,info,out,st); torch.cuda.synchronize()
got=out.cpu().numpy().tobytes()
shown=0
for i in range(0,len(d),bs):
    a,b=d[i:i+bs],got[i:i+bs]
    if a!=b:
        k=next(j for j in range(len(a)) if a[j]!=b[j])
        print('block',i//bs,'len',len(a),'stream len',len(streams[i//bs]),'first mismatch at',k,'expect',a[k:k+12].hex(),'got',b[k:k+12].hex())
        shown+=1
        if shown>=12: break
