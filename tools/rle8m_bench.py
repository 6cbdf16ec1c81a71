# rle8m decode on the GPU: python tools/rle8m_bench.py [size_mib] [section_bytes] [kind]
# The stream is produced on the host by the oracle's rle8m_compress (the reference's rle8m_compress is a CPU function too); the GPU
# decodes it device-resident.  CPU column: the compiled reference's rle8m_decompress when oracle/_ref is present, else the oracle.
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
dev=torch.frombuffer(bytearray(st),dtype=torch.uint8).cuda()
info=hsrle.rle8m_info(dev)
out=torch.empty(size,dtype=torch.uint8,device='cuda'); status=torch.zeros(1,dtype=torch.int32,device='cuda')
for _ in range(3): hsrle.rle8m_decompress_async(dev,info,out,status)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
n=10; e0.record()
for _ in range(n): hsrle.rle8m_decompress_async(dev,info,out,status)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/n
ok=int(status.item())==0 and torch.equal(out,src)
cpu=Reference() if os.path.exists(REF_SO) else ora
sample=min(size,256<<20)//sec*sec
sst=ora.rle8m_compress(sample//sec,host[:sample])
t0=time.perf_counter(); got=cpu.rle8m_decompress(sst,sample); tcpu=time.perf_counter()-t0
print('rle8m decode: %d MiB, %d sections of %d B, ratio %.4f | GPU %.3f ms = %.0f GiB/s (%.1f %% of 8 TB/s on C+U) | CPU %s 1 thread %.2f GiB/s | host compress %.2f GiB/s | %s'%(
  size>>20,sections,sec,len(st)/size,ms,size/2**30/(ms*1e-3),(size+len(st))/(ms*1e-3)/8e12*100,'reference' if os.path.exists(REF_SO) else 'oracle',sample/2**30/tcpu,size/2**30/tc,'ok' if ok and got==host[:sample] else 'FAIL'))
