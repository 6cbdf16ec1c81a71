# in-kernel phase stamps (diagnostic build): HSRLE_LIB=variants/libhsrle_stamps.so python tools/probe_kernel_stamps.py
import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'hypersonic-rle-kit_amd/python')
import torch, hsrle
size=int(float(sys.argv[1])*(1<<20)) if len(sys.argv)>1 else 1024<<20
bs=int(sys.argv[2]) if len(sys.argv)>2 else 4096
codec=sys.argv[3] if len(sys.argv)>3 else 'rle8_packed_multi'
S={'8':1,'16':2,'24':3,'32':4,'48':6,'64':8,'128':16}[codec.split('_')[0][3:]]
kind=int(sys.argv[4]) if len(sys.argv)>4 else 0
src=hsrle.synth(kind,S,2,size)
cont,info=hsrle.compress(codec,src,block_size=bs)
out=torch.empty(size,dtype=torch.uint8,device='cuda'); st=torch.zeros(256,dtype=torch.int32,device='cuda')
hsrle.decompress_async(cont,info,out,st); torch.cuda.synchronize()
st.zero_()
hsrle.decompress_async(cont,info,out,st); torch.cuda.synchronize()
d=st[16:].view(torch.int64).cpu().tolist()+[0]*12
if len(d)>16 and d[12]: print('topup: exchange %.0f land(wait+ds_write) %.0f issue loads %.0f | flush: exchange %.0f reads+stores %.0f (cycles per round)'%tuple(x/d[4] for x in d[12:17]))
tI,tD,tF,tL,nR,nIt,nW=d[:7]
print('lane0 events: partial rows',tI>>40,'literal-starved',tF>>40,'header-starved',tL>>40)
tI&=(1<<40)-1; tF&=(1<<40)-1; tL&=(1<<40)-1
tot=tI+tD+tF+tL
print('waves',nW,'rounds/wave',nR/nW,'iters/round',nIt/nR)
print('cycles per round: issue %.0f decode %.0f flush %.0f land %.0f total %.0f'%(tI/nR,tD/nR,tF/nR,tL/nR,tot/nR))
if len(d)>11 and d[11]:
    print('wave-level iterations/round %.2f ; per lane-iteration cycles: parse %.0f lit %.0f run %.0f (lane-iterations %d)'%(d[7]/nR, d[8]/d[11], d[9]/d[11], d[10]/d[11], d[11]))
print('ok', int(st[0].item())==0 and torch.equal(out,src))
