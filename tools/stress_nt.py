"""Randomised stress of the persistent NT kernels (gemm_nt_v7 / gemm_nt_v9): shapes, forms, cu_margin and tile scheduler drawn at random,
every product checked against fp32.  Run from the repo root on a GPU box: python tools/stress_nt.py [seed]."""
import ctypes as C, os, random, sys, torch
sys.path.insert(0, "/root/repo")
from linnaeus_amd import _lib as L
from tests.test_gpu_gemm import run_nt
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(160):
    kind = rnd.choice(["v7", "v9"])
    margin = rnd.choice([0, 0, 1, 7, 100, 200, 248])
    L.check(L.lib().lnx_set_cu_margin(margin), "m")
    os.environ["LNX_NT_V7"] = "1" if kind == "v7" else "0"
    os.environ["LNX_NT_V9"] = "1" if kind == "v9" else "0"
    os.environ["LNX_TILE_SCHED"] = rnd.choice(["atomic", "atomic", "static"])
    M = rnd.randrange(1024, 30000)
    N = rnd.choice([256, 512, 768, 1280]) if kind == "v9" else 64 * rnd.randrange(1, 24)
    K = rnd.choice([256, 384, 512, 640, 1152, 2048]) if kind == "v9" else rnd.choice([384, 448, 512, 1152, 1536])
    form = rnd.choice(["plain", "bias", "res", "aux"])
    g = torch.Generator(device="cuda").manual_seed(it)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    z = A.float() @ W.float().t()
    if form == "plain":
        out, _ = run_nt(A, W, L.BF16, False); ref = z
    elif form == "bias":
        out, _ = run_nt(A, W, L.BF16, False, bias=b); ref = z + b
    elif form == "res":
        res = torch.randn(M, N, device="cuda", generator=g)
        out, _ = run_nt(A, W, L.BF16, True, bias=b, res=res); ref = res + z + b
    else:
        aux = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        out, _ = run_nt(A, W, L.BF16, False, act=L.ACT_MUL_AUX, aux=aux); ref = z * aux.float()
    got = L.lib().lnx_last_nt_kernel()
    err = (out.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    ok = err < 1e-2
    if not ok or it % 40 == 0:
        print(it, kind, "kernel", got, "margin", margin, os.environ["LNX_TILE_SCHED"], M, N, K, form, "err %.2e" % err, "OK" if ok else "BAD", flush=True)
    bad += not ok
L.check(L.lib().lnx_set_cu_margin(0), "m")
print("bad cases:", bad)
sys.exit(1 if bad else 0)
