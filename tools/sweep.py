# Full codec sweep (BASELINE config 5 shape): python tools/sweep.py [size_mib] [block] -> markdown table on stdout
import sys, os, json
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import CODECS
size=(int(sys.argv[1]) if len(sys.argv)>1 else 1024)<<20
bs=int(sys.argv[2]) if len(sys.argv)>2 else 4096
kinds=[('runs',0)] + ([('video',1)] if len(sys.argv)>3 else [])
def bench(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n/1e3
rows=[]
print('| codec | data | ratio | encode GiB/s | decode GiB/s | decode % of 8 TB/s (C+U) | round trip |'); print('|---|---|---:|---:|---:|---:|---|')
for kname,kind in kinds:
    cache={}
    for c in CODECS:
        if c.S not in cache: cache[c.S]=hsrle.synth(kind,c.S,5,size)
        src=cache[c.S]
        dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda'); ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
        hsrle.compress_async(c.key,src,dst,bs,workspace=ws); torch.cuda.synchronize()
        info=hsrle.container_info(dst)
        out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(16,dtype=torch.int32,device='cuda')
        td=bench(lambda: hsrle.decompress_async(dst,info,out,st)); te=bench(lambda: hsrle.compress_async(c.key,src,dst,bs,workspace=ws),2)
        ok=int(st[0].item())==0 and torch.equal(out,src)
        print('| %s | %s | %.4f | %.0f | %.0f | %.1f | %s |'%(c.key,kname,info.totalSize/size,size/te/2**30,size/td/2**30,(size+info.totalSize)/td/8e10,'ok' if ok else 'FAIL'),flush=True)
        del dst,ws,out
