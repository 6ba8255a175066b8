import os, sys, torch, collections
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import _lib
h=_lib.debug_lib()
B=int(os.environ.get("KB_B","512")); S=32
xn=torch.randn(B,65,192,device="cuda").bfloat16(); from vitpe import kernels as K
w=K.pack_qkv_weights(torch.randn(576,192,device="cuda")*0.1, torch.bfloat16, 6); out=torch.empty_like(xn)
nwg=(B+1)//2
cen=torch.zeros(nwg*16*S,dtype=torch.int64,device="cuda")
for _ in range(3):
    h.vitpe_debug_attn_census(xn.data_ptr(), w.data_ptr(), out.data_ptr(), B, cen.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c=cen.cpu().numpy().reshape(nwg,16,S)[:, :12, :]
t0=c[:,:,0].min()
names={0:"start",1:"staged",2:"bar0"}
for p in range(3):
    names.update({3+4*p:f"p{p}.proj",4+4*p:f"p{p}.bar1",5+4*p:f"p{p}.core",6+4*p:f"p{p}.bar2"})
names[30]="end"
order=sorted(names)
rel=c - c[:,:,0:1].min(axis=1, keepdims=True)
print("phase           " + " ".join(f"  w{w:02d}" for w in range(12)) + "   (median cycles since WG start)")
for s_ in order:
    print(f"{names[s_]:14s} " + " ".join(f"{int(np.median(rel[:,w,s_])):5d}" for w in range(12)))
print("kernel span cycles:", int(c[:,:,30].max()-t0), " WG start spread:", int(c[:,:,0].min(axis=1).max()-t0))
