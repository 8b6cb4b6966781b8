"""fp8 (per-tensor scale) and MXFP8 (block scales in the MFMA) vs bf16 NT GEMM at the model's shapes (sm RoPE stages, xl RoPE stage 2, two cubes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from linnaeus_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [("xl b128 qkv", 25600, 3072, 1024), ("xl b128 fc1", 25600, 4096, 1024), ("xl b128 fc2", 25600, 1024, 4096), ("xl b128 proj", 25600, 1024, 1024), ("sm r0.qkv", 50176, 1152, 384), ("sm r0.fc1", 50176, 1536, 384), ("sm r0.fc2", 50176, 384, 1536), ("sm r1.fc1", 12544, 3072, 768), ("sm r1.fc2", 12544, 768, 3072),
          ("xl r0.qkv", 50176, 3072, 1024), ("xl r0.fc1", 50176, 4096, 1024), ("xl r0.fc2", 50176, 1024, 4096), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a8, sa = ops.quantize_fp8(a)
    w8, sw = ops.quantize_fp8(w)
    tb = t(lambda: ops.gemm_nt(a, w, out))
    t8 = t(lambda: ops.gemm_nt_fp8(a8, sa, w8, sw, out))
    tq = t(lambda: ops.quantize_fp8(a))
    am, sam = ops.quantize_mxfp8(a)
    wm, swm = ops.quantize_mxfp8(w)
    os.environ["LNX_FP8_X8"] = "0"  # the 256x128 MX kernel
    tm0 = t(lambda: ops.gemm_nt_mxfp8(am, sam, wm, swm, out))
    os.environ.pop("LNX_FP8_X8")     # default: the 256x256 one where it applies
    tm = t(lambda: ops.gemm_nt_mxfp8(am, sam, wm, swm, out))
    tqm = t(lambda: ops.quantize_mxfp8(a))
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:5d}: bf16 {tb:7.1f} us ({fl / tb / 1e6:6.0f} TF/s)  fp8 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TF/s) x{tb / t8:4.2f}"
          f"  mxfp8 256x128 {tm0:7.1f} us ({fl / tm0 / 1e6:6.0f} TF/s)  default {tm:7.1f} us ({fl / tm / 1e6:6.0f} TF/s) x{tb / tm:4.2f}   amax+quantise(A) {tq:6.1f} us  mx quantise(A) {tqm:6.1f} us")
