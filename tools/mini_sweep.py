# quick decode/encode check on a few codecs x {runs, video, random}: python tools/mini_sweep.py [size_mib]
import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
size=(int(sys.argv[1]) if len(sys.argv)>1 else 1024)<<20
bs=4096
keys=(sys.argv[2].split(',') if len(sys.argv)>2 else ['rle8_multi','rle8_packed_multi','rle8_3symlut','rle8_single','rle8_packed_single','rle16_sym','rle24_byte_packed','rle64_3symlut_byte','rle128_sym'])
def bench(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n/1e3
g=torch.Generator(device='cuda'); g.manual_seed(7)
rnd=torch.randint(0,256,(size,),dtype=torch.uint8,device='cuda',generator=g)
for kname,kind in (('runs',0),('video',1),('random',-1)):
    for k in keys:
        c=CODEC_BY_KEY[k]
        src=rnd if kind<0 else hsrle.synth(kind,c.S,5,size)
        dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda'); ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
        hsrle.compress_async(k,src,dst,bs,workspace=ws); torch.cuda.synchronize()
        info=hsrle.container_info(dst)
        out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(16,dtype=torch.int32,device='cuda')
        td=bench(lambda: hsrle.decompress_async(dst,info,out,st)); te=bench(lambda: hsrle.compress_async(k,src,dst,bs,workspace=ws),2)
        ok=int(st[0].item())==0 and torch.equal(out,src)
        print('%-22s %-6s ratio %.4f  enc %5.0f  dec %5.0f GiB/s  %.1f%%  %s'%(k,kname,info.totalSize/size,size/te/2**30,size/td/2**30,(size+info.totalSize)/td/8e10,'ok' if ok else 'FAIL'),flush=True)
        del dst,ws,out
