import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(it)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
for D,H in ((192,6),(96,3)):
    for B in (128,256,512,1024):
        xn=torch.randn(B,65,D,device="cuda").bfloat16(); w=(torch.randn(3*D,D,device="cuda")*0.1).bfloat16(); out=torch.empty_like(xn)
        pe=K.PETables("none",8)
        print(f"D={D} B={B} attn_fwd none: {t(lambda: K.fused_attention_fwd(xn,w,H,pe,out=out)):.1f} us")
