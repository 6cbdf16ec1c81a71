"""Encode time of the 88 MB video-shaped frame and of a 64 MiB run-distributed buffer in 4 KiB blocks (small containers: the split encode):  python tools/frame_enc_time.py <codec>"""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R+'/hypersonic-rle-kit_amd/python'); sys.path.insert(0,R+'/tests')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
key=sys.argv[1]; S=CODEC_BY_KEY[key].S
for kind,size in ((1,88473600),(0,64<<20)):
    src=hsrle.synth(kind,S,2,size,device="cuda")
    dst=torch.empty(hsrle.container_bound(size,4096),dtype=torch.uint8,device="cuda"); ws=torch.empty(hsrle.workspace_size(size,4096,codec=key),dtype=torch.uint8,device="cuda")
    for _ in range(3): hsrle.compress_async(key,src,dst,4096,workspace=ws)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): hsrle.compress_async(key,src,dst,4096,workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print(key,"kind",kind,"bytes",size,"encode us",round(e0.elapsed_time(e1)/20*1e3,1))
