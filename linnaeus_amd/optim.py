"""Optimizer + step glue on device: `FusedAdamW` is a `torch.optim.Optimizer` whose
`step()` is one HIP launch for the whole model, three with clipping -- the sum of squares of every gradient (per-workgroup
partials + a fixed-order fold: bit-identical on every data-parallel rank), then a multi-tensor AdamW update with the
global-norm clip folded in (csrc/optim.hip) -- instead of `clip_grad_norm_` (the reference makes three
gradient passes with host syncs, train.py:282-308) followed by `torch.optim.AdamW.step`.

Same update rule and `param_groups` / `state_dict` layout as `torch.optim.AdamW` (per-group `lr`, `betas`, `eps`,
`weight_decay`; per-parameter `step`, `exp_avg`, `exp_avg_sq`), so LR schedulers and the reference's parameter-group
builder (optimizers/build.py) work unchanged.  GPU fp32 parameters only.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib as L


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > L.ADAMW_MAX_GROUPS:
            raise ValueError(f"at most {L.ADAMW_MAX_GROUPS} parameter groups")
        self.max_grad_norm = max_grad_norm
        self._table = None
        self._key = None
        self._sumsq = None
        self._sumsq_ws = None

    # ------------------------------------------------------------------ descriptor table
    def _build(self, items):
        dev = items[0][1].device
        lib = L.lib()
        arr = (L.AdamWDesc * len(items))()
        blk = 0
        for i, (gi, p) in enumerate(items):  # gi: hyper-parameter slot = (param group, step count) combination
            st = self.state[p]
            arr[i].p, arr[i].m, arr[i].v, arr[i].g = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.grad.data_ptr()
            arr[i].n, arr[i].group, arr[i].block_start = p.numel(), gi, blk
            blk += lib.lnx_adamw_blocks(C.c_int64(p.numel()))
        host = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy())
        self._table = (host.to(dev), len(items), blk)
        if self._sumsq is None or self._sumsq.device != dev:
            self._sumsq = torch.zeros(1, device=dev, dtype=torch.float32)
        if self._sumsq_ws is None or self._sumsq_ws.device != dev or self._sumsq_ws.numel() < blk:
            self._sumsq_ws = torch.empty(blk, device=dev, dtype=torch.float32)  # one partial per workgroup (lnx_grad_sumsq: fixed-order fold)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    raise L.LnxError("FusedAdamW needs contiguous fp32 parameters and gradients on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] = int(st["step"]) + 1  # int() also accepts the tensor a torch.optim.AdamW state_dict carries
                items.append((gi, p))
        if not items:
            return loss
        # torch.optim.AdamW bias-corrects PER PARAMETER (its own step count).  Parameters of one group normally share a
        # step count, but one that was frozen for a while, had no gradient on some steps or came from a checkpoint with
        # mixed steps does not: every distinct (group, step) pair gets its own hyper-parameter slot.
        slots = {}
        for gi, p in items:
            slots.setdefault((gi, int(self.state[p]["step"])), len(slots))
        if len(slots) > L.ADAMW_MAX_GROUPS:
            raise L.LnxError(f"FusedAdamW: {len(slots)} distinct (parameter group, step count) combinations, at most {L.ADAMW_MAX_GROUPS} are supported")
        items = [(slots[(gi, int(self.state[p]["step"]))], p) for gi, p in items]
        key = tuple((si, p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for si, p in items)
        if key != self._key:
            self._build(items)
            self._key = key
        h = L.AdamWHyper()
        h.ngroups = len(slots)
        for (gi, t), si in slots.items():
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            h.lr[si], h.beta1[si], h.beta2[si], h.eps[si], h.weight_decay[si] = group["lr"], b1, b2, group["eps"], group["weight_decay"]
            h.bias_c1[si], h.bias_c2[si] = 1.0 - b1 ** float(t), 1.0 - b2 ** float(t)
            h.omb1[si], h.omb2[si] = 1.0 - b1, 1.0 - b2
        table, n, blocks = self._table
        lib = L.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            L.check(lib.lnx_grad_sumsq(C.c_void_p(table.data_ptr()), n, blocks, C.c_void_p(self._sumsq.data_ptr()), C.c_void_p(self._sumsq_ws.data_ptr()), stream),
                    "lnx_grad_sumsq")
        L.check(lib.lnx_adamw_step(C.c_void_p(table.data_ptr()), n, blocks, C.byref(h), C.c_void_p(self._sumsq.data_ptr()) if clip else None,
                                   C.c_float(self.max_grad_norm if clip else 0.0), stream), "lnx_adamw_step")
        return loss

    def grad_norm(self) -> Optional[torch.Tensor]:
        """total L2 norm of the gradients seen by the last clipped step (device scalar, no sync); None without clipping"""
        if self._sumsq is None or self.max_grad_norm is None:
            return None
        return self._sumsq.sqrt()[0]


# ----------------------------------------------------------------------------------------------------------------------
# Muon: SGD-momentum whose 2-D update is orthogonalised by five Newton-Schulz iterations in bf16
# (optimizers/muon.py:27-65, 67-160).  The 15 matrix products per parameter run on lnx_gemm_nt (bf16 MFMA, fp32
# accumulate): A = X X^T, B = c A A + b A and X' = B X + a X are each ONE launch (the affine terms ride in the GEMM
# epilogue as the per-column scale and the fp32 residual), where the reference issues a matmul plus elementwise
# bf16 passes per term.  Same constructor, param_groups and state ('momentum_buffer') as the reference's Muon.
# ----------------------------------------------------------------------------------------------------------------------
def zeropower_via_newtonschulz5(G: torch.Tensor, steps: int = 5) -> torch.Tensor:
    """Orthogonalise a 2-D gradient (quintic Newton-Schulz, coefficients of optimizers/muon.py:41).  bf16 in / out."""
    from . import ops

    assert G.ndim == 2, "2-D matrices (4-D conv weights are flattened by the optimizer)"
    if not G.is_cuda:
        raise L.LnxError("linnaeus_amd.optim.Muon runs on the HIP GEMM kernels: parameters must be on the GPU")
    a, b, c = 3.4445, -4.7750, 2.0315
    X = G.bfloat16()
    tall = G.size(0) > G.size(1)
    if tall:
        X = X.t()
    m, n = X.shape
    mp, npad = (m + 7) // 8 * 8, (n + 7) // 8 * 8   # GEMM K granularity; zero rows / columns do not change any product
    X = X / (X.float().norm() + 1e-7).to(torch.bfloat16)
    if (mp, npad) != (m, n):
        X = torch.nn.functional.pad(X, (0, npad - n, 0, mp - m))
    X = X.contiguous()
    dev = X.device
    cvec = torch.full((mp,), c, device=dev, dtype=torch.float32)
    ones = None
    A = torch.empty(mp, mp, device=dev, dtype=torch.bfloat16)
    Bm = torch.empty(mp, mp, device=dev, dtype=torch.bfloat16)
    Xn = torch.empty(mp, npad, device=dev, dtype=torch.bfloat16)
    for _ in range(steps):
        XT = X.t().contiguous()                                   # [n, m]: the "W" operand of B . X
        ops.gemm_nt(X, X, A)                                      # A = X X^T
        ops.gemm_nt(A, A, Bm, gamma=cvec, res=A.float() * b)      # B = c A A + b A   (A symmetric: A A^T = A A)
        ops.gemm_nt(Bm, XT, Xn, res=X.float() * a)                # X' = B X + a X
        X, Xn = Xn, X
    X = X[:m, :n]
    return (X.t() if tall else X).contiguous()


class Muon(torch.optim.Optimizer):
    """optimizers/muon.py:67-160 on the HIP kernels (2-D parameters, or 4-D conv weights flattened to [out, -1])."""

    def __init__(self, params, lr=0.02, weight_decay=0.01, momentum=0.95, nesterov=True, ns_steps=5, strict=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= momentum < 1.0:
            raise ValueError(f"Invalid momentum value: {momentum}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        params = list(params)
        if strict:
            for p in (q for g in params for q in (g["params"] if isinstance(g, dict) else [g])):
                if p.dim() not in (2, 4):
                    raise ValueError(f"Muon optimizer requires 2D or 4D parameters, got shape {p.shape}")
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, momentum=momentum, nesterov=nesterov, ns_steps=ns_steps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, wd, mom = group["lr"], group["weight_decay"], group["momentum"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if wd != 0:
                    p.mul_(1 - lr * wd)
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(g)
                buf = st["momentum_buffer"]
                buf.lerp_(g, 1 - mom)
                g = g.lerp_(buf, mom) if group["nesterov"] else buf.clone()
                g2 = g.view(g.size(0), -1) if g.ndim == 4 else g
                o = zeropower_via_newtonschulz5(g2, steps=group["ns_steps"])
                if p.dim() == 4:
                    scaling = max(1, p.size(0) / (p.size(1) * p.size(2) * p.size(3))) ** 0.5
                else:
                    scaling = max(1, p.size(-2) / p.size(-1)) ** 0.5
                p.add_(o.view_as(p).to(p.dtype), alpha=-lr * scaling)
        return loss
