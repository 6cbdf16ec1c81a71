# rle8m decode on the GPU: python tools/rle8m_bench.py [size_mib] [section_bytes] [kind]
# Encode and decode device resident (hsrle_rle8m_compress_dev_async / hsrle_rle8m_decompress_dev_async); the stream is compared with the
# oracle's rle8m_compress of the same input.  CPU columns: the compiled reference when oracle/_ref is present, else the oracle (1 thread).
import sys, os, time, ctypes
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import numpy as np, torch, hsrle
from hsrle_testlib import Oracle, Reference, REF_SO
size=(int(sys.argv[1]) if len(sys.argv)>1 else 1024)<<20
sec=int(sys.argv[2]) if len(sys.argv)>2 else 4096
kind=int(sys.argv[3]) if len(sys.argv)>3 else 0
ora=Oracle()
src=hsrle.synth(kind,1,7,size)
host=src.cpu().numpy().tobytes()
sections=size//sec
t0=time.perf_counter(); st=ora.rle8m_compress(sections,host); tc=time.perf_counter()-t0
assert st is not None, 'rle8m_compress gave up (the stream outgrew its bound)'
dst=torch.empty(hsrle.rle8m_bounds(sections,size),dtype=torch.uint8,device='cuda'); ws=torch.empty(hsrle.rle8m_workspace_size(size,sections),dtype=torch.uint8,device='cuda')
status=torch.zeros(1,dtype=torch.int32,device='cuda')
def ev(fn,n):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
mse=ev(lambda: hsrle.rle8m_compress_async(src,sections,dst,ws,status),5)
info=hsrle.rle8m_info(dst)
same=int(status.item())==0 and dst[:info.compressedSize].cpu().numpy().tobytes()==st
dev=dst[:info.compressedSize]
out=torch.empty(size,dtype=torch.uint8,device='cuda')
ms=ev(lambda: hsrle.rle8m_decompress_async(dev,info,out,status),10)
ok=int(status.item())==0 and torch.equal(out,src)
cpu=Reference() if os.path.exists(REF_SO) else ora
sample=min(size,256<<20)//sec*sec
t0=time.perf_counter(); sst=cpu.rle8m_compress(sample//sec,host[:sample]); tce=time.perf_counter()-t0
t0=time.perf_counter(); got=cpu.rle8m_decompress(sst,sample); tcpu=time.perf_counter()-t0
print('rle8m: %d MiB, %d sections of %d B, ratio %.4f | GPU encode %.3f ms = %.0f GiB/s, decode %.3f ms = %.0f GiB/s (%.1f %% of 8 TB/s on C+U) | CPU %s 1 thread: encode %.2f, decode %.2f GiB/s | %s'%(
  size>>20,sections,sec,len(st)/size,mse,size/2**30/(mse*1e-3),ms,size/2**30/(ms*1e-3),(size+len(st))/(ms*1e-3)/8e12*100,'reference' if os.path.exists(REF_SO) else 'oracle',sample/2**30/tce,sample/2**30/tcpu,'ok' if ok and same and got==host[:sample] else 'FAIL'))
