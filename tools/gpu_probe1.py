import sys, time
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import *
O=Oracle()
print(hsrle.lib().hsrle_version(), torch.cuda.get_device_name(0))
import random
rng=random.Random(1)
d=mixed_runs(rng, 20000)
for key in ['rle8_packed_multi','rle8_multi','rle8_3symlut','rle64_3symlut_byte','rle24_sym_packed','rle128_byte_packed','rle8_single']:
    c=CODEC_BY_KEY[key]
    src=torch.frombuffer(bytearray(d),dtype=torch.uint8).cuda()
    cont,info=hsrle.compress(key,src,block_size=512)
    ci,streams=hsrle.split_container(cont.cpu().numpy().tobytes())
    bad=sum(1 for i,s in enumerate(streams) if s!=O.compress(c,d[i*512:(i+1)*512]))
    out=hsrle.decompress(cont)
    print(key,'blocks',len(streams),'bad streams',bad,'roundtrip',out.cpu().numpy().tobytes()==d)
# perf probe
for size in [1<<28]:
    src=hsrle.synth(0,1,2,size)
    torch.cuda.synchronize()
    for bs in [4096, 16384]:
        cont,info=hsrle.compress('rle8_packed_multi',src,block_size=bs)
        out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(1,dtype=torch.int32,device='cuda')
        dst=torch.empty(hsrle.container_bound(size,bs),dtype=torch.uint8,device='cuda')
        for name,fn in [('dec',lambda: hsrle.decompress_async(cont,info,out,st)),('enc',lambda: hsrle.compress_async('rle8_packed_multi',src,dst,bs))]:
            fn(); torch.cuda.synchronize()
            t0=time.time()
            for _ in range(5): fn()
            torch.cuda.synchronize(); dt=(time.time()-t0)/5
            print(name,'bs',bs,'ratio %.3f'%(info.totalSize/size),'%.2f ms'%(dt*1e3),'%.1f GiB/s'%(size/dt/2**30), 'alg GB/s %.0f'%((size+info.totalSize)/dt/1e9))
        print('ok', int(st.item()), torch.equal(out,src))
