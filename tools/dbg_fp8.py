import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops
gen = torch.Generator().manual_seed(4096 + 1152 + 768)
x = (torch.randn(4096, 768, generator=gen) * 3).cuda().bfloat16()
x8, sx = ops.quantize_fp8(x)
amax = x.float().abs().max()
v = x.float() * (448.0 / amax)
ref8 = v.clamp(-448, 448).to(torch.float8_e4m3fn)
d = (x8.view(torch.uint8).int() - ref8.view(torch.uint8).int())
idx = d.nonzero()
print("mismatch", idx.shape[0], "of", d.numel(), "amax", amax.item(), "sx*448", sx.item() * 448)
for i in idx[:8]:
    r, c = i.tolist()
    print(r, c, "x", x[r, c].item(), "v", v[r, c].item(), "mine", x8[r, c].float().item(), "torch", ref8[r, c].float().item())
print("cols of mismatches mod 16:", torch.bincount(idx[:, 1] % 16, minlength=16).tolist())
