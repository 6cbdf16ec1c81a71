import sys, os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R+'/hypersonic-rle-kit_amd/python'); sys.path.insert(0,R+'/tests')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
key=sys.argv[1]; S=CODEC_BY_KEY[key].S
for mib in (576, 640, 704, 768, 896):
    for kind in (0,1):
        size=mib<<20
        src=hsrle.synth(kind,S,2,size,device="cuda")
        dst=torch.empty(hsrle.container_bound(size,4096),dtype=torch.uint8,device="cuda"); ws=torch.empty(hsrle.workspace_size(size,4096),dtype=torch.uint8,device="cuda")
        for _ in range(3): hsrle.compress_async(key,src,dst,4096,workspace=ws)
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): hsrle.compress_async(key,src,dst,4096,workspace=ws)
        e1.record(); torch.cuda.synchronize()
        print(os.environ.get("HSRLE_RUNLIST","-"),key,"MiB",mib,"kind",kind,"encode us",round(e0.elapsed_time(e1)/10*1e3,1),"GiB/s",round(mib/1024/(e0.elapsed_time(e1)/10/1e3),1),flush=True)
        del src,dst,ws
