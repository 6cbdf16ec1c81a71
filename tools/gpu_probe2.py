import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev0=torch.cuda.Event(enable_timing=True); ev1=torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)/n/1e3
for size in [1<<30, 1<<33]:
    src=hsrle.synth(0,1,2,size); torch.cuda.synchronize()
    for bs in [1024, 2048, 4096, 8192]:
        dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda')
        ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
        hsrle.compress_async('rle8_packed_multi',src,dst,bs,workspace=ws); torch.cuda.synchronize()
        info=hsrle.container_info(dst)
        out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(1,dtype=torch.int32,device='cuda')
        td=bench(lambda: hsrle.decompress_async(dst,info,out,st))
        te=bench(lambda: hsrle.compress_async('rle8_packed_multi',src,dst,bs,workspace=ws), 3)
        ok = int(st.item())==0 and torch.equal(out,src)
        print('size %d MiB bs %d ratio %.4f | dec %.2f ms %.0f GiB/s (alg %.0f GB/s) | enc %.2f ms %.0f GiB/s | ok %s'%(size>>20,bs,info.totalSize/size,td*1e3,size/td/2**30,(size+info.totalSize)/td/1e9,te*1e3,size/te/2**30,ok), flush=True)
        del dst, ws, out
