"""Encode time of one codec's (default rle8_packed_multi) containers of blocks above 4 KiB, small to large (where does the windowed position-parallel encoder overtake the split / ring paths?):
python tools/ppw_threshold.py [codec]  (run once per library: HSRLE_LIB)"""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R+'/hypersonic-rle-kit_amd/python')
import torch, hsrle
sys.path.insert(0,R+"/tests")
from hsrle_testlib import CODEC_BY_KEY
key=sys.argv[1] if len(sys.argv)>1 else "rle8_packed_multi"; S=CODEC_BY_KEY[key].S
for size in (8<<20, 16<<20, 32<<20, 88473600, 256<<20):
    for B in (8192, 65536):
        for kind in (0, 1):
            src=hsrle.synth(kind,S,2,size,device="cuda")
            dst=torch.empty(hsrle.container_bound(size,B),dtype=torch.uint8,device="cuda"); ws=torch.empty(hsrle.workspace_size(size,B,codec=key),dtype=torch.uint8,device="cuda")
            for _ in range(3): hsrle.compress_async(key,src,dst,B,workspace=ws)
            torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(20): hsrle.compress_async(key,src,dst,B,workspace=ws)
            e1.record(); torch.cuda.synchronize()
            print(os.environ.get("HSRLE_LIB","default").split("/")[-1],key,"bytes",size,"B",B,"blocks",(size+B-1)//B,"kind",kind,"path",hsrle.lib().hsrle_encode_path(hsrle.codec_id(key),size,B),"encode us",round(e0.elapsed_time(e1)/20*1e3,1),flush=True)
