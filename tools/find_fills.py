import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: arch="sm"; img=224
cfg, model = bench.make_model(A)
model = model.cuda(); model.set_compute_dtype("bf16"); model.train(); model.grad_mode = "direct"
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05, fused=True)
B=32
x = torch.rand(B,3,224,224,device="cuda"); meta=torch.rand(B,5,device="cuda")
tg = {t: torch.randint(1,c,(B,),device="cuda") for t,c in bench.TASKS}
def step():
    model.zero_grad(set_to_none=True)
    out = model(x, meta)
    loss = out[bench.TASKS[0][0]].new_zeros(())
    for t,_ in bench.TASKS: loss = loss + F.cross_entropy(out[t], tg[t])
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
import collections
cnt=collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_","aten::zero_","aten::zeros","aten::zeros_like","aten::new_zeros","aten::ones_like"):
        cnt[(e.name, str(e.input_shapes)[:60])]+=1
for k,v in cnt.most_common(20): print(v,k)
