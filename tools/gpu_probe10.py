# bimodal decode time: which buffer's placement matters?  python tools/gpu_probe10.py
import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=8<<30; bs=4096
def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
src=hsrle.synth(0,1,2,size)
dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda'); ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
hsrle.compress_async('rle8_packed_multi',src,dst,bs,workspace=ws); torch.cuda.synchronize()
info=hsrle.container_info(dst)
cont=dst[:info.totalSize]
st=torch.zeros(16,dtype=torch.int32,device='cuda')
outs=[torch.empty(size+(k<<20),dtype=torch.uint8,device='cuda') for k in range(4)]
for rep in range(2):
    for k,o in enumerate(outs):
        for off in (0, 4096, 65536, 1<<20):
            if off+size<=o.numel():
                t=bench(lambda: hsrle.decompress_async(cont,info,o[off:off+size],st))
                print('out buffer %d (ptr %x) offset %7d: %.3f ms'%(k,o.data_ptr(),off,t),flush=True)
# container copies at different offsets
big=torch.empty(info.totalSize+(8<<20),dtype=torch.uint8,device='cuda')
for off in (0,64,4096,1<<20,(1<<20)+2048):
    c2=big[off:off+info.totalSize]; c2.copy_(cont)
    t=bench(lambda: hsrle.decompress_async(c2,info,outs[0][:size],st))
    print('container copy at offset %8d (ptr %x): %.3f ms'%(off,c2.data_ptr(),t),flush=True)
