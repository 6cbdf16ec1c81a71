import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R+'/hypersonic-rle-kit_amd/python'); sys.path.insert(0,R+'/tests')
import torch, hsrle
from hsrle_testlib import CODEC_BY_KEY
key=sys.argv[1] if len(sys.argv)>1 else "rle64_3symlut_byte"; S=CODEC_BY_KEY[key].S
size=88473600
src=hsrle.synth(1,S,2,size,device="cuda")
dst=torch.empty(hsrle.container_bound(size,4096),dtype=torch.uint8,device="cuda"); ws=torch.empty(hsrle.workspace_size(size,4096),dtype=torch.uint8,device="cuda")
for _ in range(10): hsrle.compress_async(key,src,dst,4096,workspace=ws)
torch.cuda.synchronize()
