import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-rpe-rope_amd"))
from vitpe import kernels as K, _lib as L
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(it)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
M=33280; D=192; hid=768; T=torch.bfloat16
x=torch.randn(M,D,device="cuda").to(T); w1=(torch.randn(hid,D,device="cuda")*0.1).to(T); b1=torch.zeros(hid,device="cuda")
h=torch.empty(M,hid,device="cuda",dtype=T); u=torch.empty_like(h)
print("fc1 bias only (write 51MB):", t(lambda: K.linear(x,w1,b1,epi=L.EPI_BIAS,out=h)))
print("fc1 bias+gelu (write 102MB):", t(lambda: K.linear(x,w1,b1,epi=L.EPI_BIAS_GELU,u=u,out=h)))
print("copy 51MB->51MB torch:", t(lambda: u.copy_(h)))
y=torch.empty(M,D,device="cuda",dtype=T)
print("copy 12.8MB torch:", t(lambda: y.copy_(x)))
print("gelu torch 51MB:", t(lambda: torch.nn.functional.gelu(h)))
