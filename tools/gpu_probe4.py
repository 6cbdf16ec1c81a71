# correctness on a few codecs + decode/encode perf for the library selected by HSRLE_LIB
import sys, time, os, random
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import *
O=Oracle()
print('LIB', os.environ.get('HSRLE_LIB','default'))
rng=random.Random(1)
d=mixed_runs(rng, 30000)+bytes(rng.randrange(256) for _ in range(5000))+mixed_runs(rng,5000)
allok=True
for key in ['rle8_packed_multi','rle8_multi','rle8_3symlut','rle8_7symlut','rle64_3symlut_byte','rle24_sym_packed','rle128_byte_packed','rle8_single','rle48_byte','rle16_7symlut_sym']:
    c=CODEC_BY_KEY[key]
    for bs in (128,512,4096):
        src=torch.frombuffer(bytearray(d),dtype=torch.uint8).cuda()
        cont,info=hsrle.compress(key,src,block_size=bs)
        out=hsrle.decompress(cont)
        ok = out.cpu().numpy().tobytes()==d
        allok &= ok
        if not ok: print('FAIL',key,bs)
print('roundtrip all ok:',allok)
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev0=torch.cuda.Event(enable_timing=True); ev1=torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)/n/1e3
sizes=[int(x)<<20 for x in (sys.argv[1].split(',') if len(sys.argv)>1 else ['1024'])]
bss=[int(x) for x in (sys.argv[2].split(',') if len(sys.argv)>2 else ['2048','4096'])]
codec=sys.argv[3] if len(sys.argv)>3 else 'rle8_packed_multi'
for size in sizes:
    src=hsrle.synth(0,1,2,size); torch.cuda.synchronize()
    for bs in bss:
        dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda')
        ws=torch.empty(hsrle.workspace_size(size,bs),dtype=torch.uint8,device='cuda')
        hsrle.compress_async(codec,src,dst,bs,workspace=ws); torch.cuda.synchronize()
        info=hsrle.container_info(dst)
        out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(1,dtype=torch.int32,device='cuda')
        td=bench(lambda: hsrle.decompress_async(dst,info,out,st))
        te=bench(lambda: hsrle.compress_async(codec,src,dst,bs,workspace=ws), 3)
        ok = int(st.item())==0 and torch.equal(out,src)
        print('size %d MiB bs %d ratio %.4f | dec %.2f ms %.0f GiB/s (alg %.0f GB/s = %.1f%% of 8TB/s) | enc %.2f ms %.0f GiB/s | ok %s'%(size>>20,bs,info.totalSize/size,td*1e3,size/td/2**30,(size+info.totalSize)/td/1e9,(size+info.totalSize)/td/8e10,te*1e3,size/te/2**30,ok), flush=True)
        del dst, ws, out
