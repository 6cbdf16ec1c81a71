# small correctness probe with status reporting (for debugging a decoder): python tools/probe_correctness.py
import sys, os, random
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import *
rng=random.Random(1)
d=mixed_runs(rng, 30000)+bytes(rng.randrange(256) for _ in range(5000))+mixed_runs(rng,5000)
keys=sys.argv[1].split(',') if len(sys.argv)>1 else [c.key for c in CODECS]
for key in keys:
    for bs in (128,512,4096):
        src=torch.frombuffer(bytearray(d),dtype=torch.uint8).cuda()
        cont,info=hsrle.compress(key,src,block_size=bs)
        out=torch.zeros(len(d),dtype=torch.uint8,device='cuda'); st=torch.zeros(64,dtype=torch.int32,device='cuda')
        hsrle.decompress_async(cont,info,out,st); torch.cuda.synchronize()
        got=out.cpu().numpy().tobytes()
        nbad=sum(1 for i in range(0,len(d),bs) if got[i:i+bs]!=d[i:i+bs])
        if nbad or int(st[0].item()): print(key,bs,'status',int(st[0].item()),'bad blocks',nbad,'of',info.blockCount, flush=True)
print('probe6 done')
