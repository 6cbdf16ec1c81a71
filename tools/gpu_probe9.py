# decode ceiling probes: constant input (one packet per block), long runs, random; python tools/gpu_probe9.py [size_mib]
import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=(int(sys.argv[1]) if len(sys.argv)>1 else 4096)<<20
bs=4096
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n/1e3
def case(name, src, key='rle8_packed_multi'):
    dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda'); ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
    hsrle.compress_async(key,src,dst,bs,workspace=ws); torch.cuda.synchronize()
    info=hsrle.container_info(dst)
    out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(16,dtype=torch.int32,device='cuda')
    td=bench(lambda: hsrle.decompress_async(dst,info,out,st)); te=bench(lambda: hsrle.compress_async(key,src,dst,bs,workspace=ws),2)
    ok=int(st[0].item())==0 and torch.equal(out,src)
    print('%-28s %-20s ratio %.4f  enc %5.0f  dec %5.0f GiB/s  (C+U)/t %.1f%% of 8TB/s  %s'%(name,key,info.totalSize/size,size/te/2**30,size/td/2**30,(size+info.totalSize)/td/8e10,'ok' if ok else 'FAIL'),flush=True)
z=torch.zeros(size,dtype=torch.uint8,device='cuda')
case('zeros',z)
# runs of exactly 64 bytes, symbols cycling
r=(torch.arange(size,device='cuda')//64%251).to(torch.uint8)
case('runs of 64',r)
r=(torch.arange(size,device='cuda')//16%251).to(torch.uint8)
case('runs of 16',r)
r=(torch.arange(size,device='cuda')//4%251).to(torch.uint8)
case('runs of 4',r)
case('video', hsrle.synth(1,1,5,size))
case('runs(8)', hsrle.synth(0,1,5,size))
case('zeros',z,'rle64_sym')
case('video', hsrle.synth(1,8,5,size),'rle64_3symlut_byte')
# plain device copy for reference
o=torch.empty_like(z); t=bench(lambda: o.copy_(z)); print('torch copy %.0f GiB/s (read+write %.1f%% of 8TB/s)'%(size/t/2**30, 2*size/t/8e10))
