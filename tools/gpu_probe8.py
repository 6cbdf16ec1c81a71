# encoder debugging: find blocks whose GPU stream differs from the oracle; print context
import sys, os, random
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
from hsrle_testlib import *
import test_gpu_parity as T
key=sys.argv[1] if len(sys.argv)>1 else 'rle8_packed_multi'; bs=int(sys.argv[2]) if len(sys.argv)>2 else 384
c=CODEC_BY_KEY[key]
O=Oracle()
data=b"".join(T._inputs(1234+CODECS.index(c),40))
src=torch.frombuffer(bytearray(data),dtype=torch.uint8).cuda()
cont,info=hsrle.compress(key,src,block_size=bs)
ci,streams=hsrle.split_container(cont.cpu().numpy().tobytes())
nbad=0
for i,s in enumerate(streams):
    blk=data[i*bs:(i+1)*bs]
    e=O.compress(c,blk)
    if s!=e:
        nbad+=1
        if nbad<=3:
            k=next((j for j in range(min(len(s),len(e))) if s[j]!=e[j]), min(len(s),len(e)))
            print('block',i,'len',len(blk),'gpu size',len(s),'oracle size',len(e),'first diff at',k)
            print(' gpu   ',s[max(0,k-12):k+24].hex())
            print(' oracle',e[max(0,k-12):k+24].hex())
            print(' input ',blk[:64].hex())
print('bad blocks',nbad,'of',len(streams))
