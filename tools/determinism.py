import os, sys, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "vit-rpe-rope_amd"))
from oracle import vit_oracle as O
from models.vit import VisionTransformer
from vitpe.engine import TrainEngine
cfg = O.VitConfig(pos_encoding="rope-axial")
model = VisionTransformer(pos_encoding="rope-axial")
with torch.no_grad():
    for n, p in model.named_parameters(): p.copy_(O.closed_form_tensor(n, tuple(p.shape), cfg))
model.cuda()
B=int(os.environ.get("KB_B","16"))
eng = TrainEngine(model, B, compute_dtype=torch.bfloat16, use_graph=False)
images, labels = O.closed_form_batch(cfg, B)
eng.images.copy_(images.cuda()); eng.labels.copy_(labels.cuda())
gs=[]; ls=[]
for i in range(4):
    eng.flat_g.zero_(); eng.forward_backward(); torch.cuda.synchronize()
    gs.append(eng.flat_g.clone()); ls.append(eng.logits.clone())
for i in range(1,4):
    d=(gs[i]-gs[0]).abs().max().item(); print("run",i,"max|dg|",d,"rel",d/gs[0].abs().max().item(),"logits equal",torch.equal(ls[i],ls[0]))
# per-parameter worst relative diff
worst=[]
for n,p in model.named_parameters():
    o=eng._off[id(p)]; a=gs[0][o:o+p.numel()]; b=gs[1][o:o+p.numel()]
    worst.append(((a-b).abs().max().item()/(a.abs().max().item()+1e-30), n))
print(sorted(worst)[-6:])
