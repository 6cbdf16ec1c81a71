# differential stress test on the GPU: python tools/gpu_stress.py [seconds] [seed]
# random inputs of many shapes x all 110 codecs (+ rle8m) x several block sizes: every block stream must equal the oracle's, the decode (plain and
# split: packet list / entry records) must equal the input, the status word must be 0; the drop-in functions' monolithic streams likewise.  Prints the first mismatches and a summary; exit code 1 on any failure.
import sys, os, time, random
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import numpy as np, torch, hsrle
from hsrle_testlib import CODECS, Oracle, fuzz_sections, mixed_runs, single_symbol_mix, FUZZ_LENGTHS
budget=float(sys.argv[1]) if len(sys.argv)>1 else 120.0
seed=int(sys.argv[2]) if len(sys.argv)>2 else 1
rng=random.Random(seed); ora=Oracle()
if os.environ.get('STRESS_KEYS'): CODECS=[c for c in CODECS if any(c.key.startswith(k) for k in os.environ['STRESS_KEYS'].split(','))]   # e.g. STRESS_KEYS=rle8_ : only these codecs
def long_runs(rng,n):
    out=bytearray()
    while len(out)<n:
        S=rng.choice([1,2,3,4,6,8,16]); sym=bytes(rng.randrange(256) for _ in range(S))
        k=rng.choice([40,200,1000,5000,70000,300000])
        out+=(sym*(k//S+2))[:k]; out+=bytes(rng.randrange(256) for _ in range(rng.choice([0,1,2,7,100,300,70000])))
    return bytes(out[:n])
def near_period(rng,n):
    # almost-periodic data: runs broken by single-byte flips every so often (stresses stretch handling across windows)
    S=rng.choice([2,3,4,6,8,16]); sym=bytes(rng.randrange(4) for _ in range(S)); b=bytearray((sym*(n//S+1))[:n])
    for _ in range(n//rng.choice([37,61,64,65,127,200])): b[rng.randrange(n)]^=rng.randrange(1,4)
    return bytes(b)
gens=[lambda r:b"".join(fuzz_sections(r,8,FUZZ_LENGTHS) for _ in range(6)), lambda r:mixed_runs(r,r.choice([3000,20000,70000])), lambda r:single_symbol_mix(r,r.choice([3000,30000])),
      lambda r:long_runs(r,r.choice([20000,400000])), lambda r:near_period(r,r.choice([5000,50000])), lambda r:bytes(r.randrange(r.choice([2,3,256])) for _ in range(r.choice([1,100,5000,40000])))]
t0=time.time(); cases=0; bad=0; blocks=0
while time.time()-t0<budget:
    data=rng.choice(gens)(rng)
    if not data: continue
    tail=rng.choice([0,0,1,5,17,127])
    if tail and len(data)>tail: data=data[:len(data)-tail]
    arr=np.frombuffer(data,dtype=np.uint8)
    src=torch.from_numpy(arr.copy()).cuda()
    for bs in rng.sample([128,256,384,1024,1536,3072,4096,4224,8192,12416,65536],3):
        for c in rng.sample(CODECS,min(10,len(CODECS))):
            cont,info=hsrle.compress(c.key,src,block_size=bs)
            _,streams=hsrle.split_container(cont.cpu().numpy().tobytes())
            exp=ora.compress_blocks(c,arr,bs)
            cases+=1; blocks+=len(exp)
            if streams!=exp:
                bad+=1
                i=next(k for k in range(len(exp)) if streams[k]!=exp[k])
                print('ENCODE MISMATCH',c.key,'block size',bs,'input len',len(data),'block',i,'gpu',len(streams[i]),'oracle',len(exp[i]),'seed',seed,flush=True)
            out=torch.zeros(len(data),dtype=torch.uint8,device='cuda'); st=torch.zeros(16,dtype=torch.int32,device='cuda')
            hsrle.decompress_async(cont,info,out,st); torch.cuda.synchronize()
            if int(st[0].item())!=0 or not torch.equal(out,src):
                bad+=1; print('DECODE MISMATCH',c.key,'block size',bs,'input len',len(data),'status',int(st[0].item()),'seed',seed,flush=True)
            # the split decode of the same container: packet list (blocks of 256 B .. 16 KiB) or entry records every `sub` bytes; garbage workspace
            if bs>=256 and rng.random()<0.5:
                sub=1 if bs<=16384 and rng.random()<0.6 else rng.choice([s_ for s_ in (128,256,512,1024) if s_<=bs and bs%s_==0])
                ws=torch.full((max(hsrle.split_workspace_size(info,None,sub),16),),0xC3,dtype=torch.uint8,device='cuda')
                out.zero_(); st.zero_()
                hsrle.decompress_split_async(cont,info,out,ws,st[:1],sub_block=sub); torch.cuda.synchronize()
                cases+=1
                if int(st[0].item())!=0 or not torch.equal(out,src):
                    bad+=1; print('SPLIT DECODE MISMATCH',c.key,'block size',bs,'sub',sub,'input len',len(data),'status',int(st[0].item()),'seed',seed,flush=True)
    # the drop-in functions (host pointers, ONE monolithic stream: many-lane encode, index + block decode): stream == oracle's, round trip
    if len(data)<=(1<<20):
        for c in rng.sample(CODECS,min(3,len(CODECS))):
            want=ora.compress(c,data)
            size,got=hsrle.call_dropin(c.cname,data,hsrle.compress_bounds(len(data)))
            cases+=1
            if size!=len(want) or got!=want:
                bad+=1; print('DROP-IN ENCODE MISMATCH',c.key,'input len',len(data),'gpu',size,'oracle',len(want),'seed',seed,flush=True)
            else:
                n2,back=hsrle.call_dropin(c.dname,want,len(data))
                if n2!=len(data) or back!=data:
                    bad+=1; print('DROP-IN DECODE MISMATCH',c.key,'input len',len(data),'gpu',n2,'seed',seed,flush=True)
    # rle8m: stream == oracle's (None where the reference gives up or overruns its output), decode == input
    for sections in rng.sample([1,2,3,7,16,64,255,1024,max(1,len(data)//rng.choice([5,64,333,4096]))],3):
        if len(data)//sections==0 or len(data)>(1<<22): continue
        want=ora.rle8m_compress(sections,data); got=hsrle.rle8m_compress_dropin(sections,data)
        cases+=1; blocks+=sections
        if got!=want:
            bad+=1; print('RLE8M ENCODE MISMATCH sections',sections,'input len',len(data),'gpu',None if got is None else len(got),'oracle',None if want is None else len(want),'seed',seed,flush=True)
        if want is not None:
            dev=torch.from_numpy(np.frombuffer(want,dtype=np.uint8).copy()).cuda(); inf=hsrle.rle8m_info(dev)
            out=torch.zeros(len(data),dtype=torch.uint8,device='cuda'); st=torch.ones(1,dtype=torch.int32,device='cuda')
            hsrle.rle8m_decompress_async(dev,inf,out,st); torch.cuda.synchronize()
            if int(st.item())!=0 or not torch.equal(out,src):
                bad+=1; print('RLE8M DECODE MISMATCH sections',sections,'input len',len(data),'status',int(st.item()),'seed',seed,flush=True)
    if bad>20: break
print('stress: %d codec x block-size cases, %d block streams compared, %d failures, %.0f s'%(cases,blocks,bad,time.time()-t0))
sys.exit(1 if bad else 0)
