"""Classification heads of the mFormerV1 path (effective reference semantics).

LinearHead            heads/linear_head.py:13-46
ConditionalClassifier heads/conditional_classifier_head.py:28-239
HierarchicalSoftmax   heads/hierarchical_softmax_head.py:28-210
configure_classification_heads  heads/utils.py:162-364

SURVEY finding F3: with a real TaxonomyTree the hierarchy matrices are named
"{parent}_{child}" while both hierarchical heads look up "{child}_{parent}", so refinement
never runs and each head returns level_classifiers[primary_task](x).  That is what these
classes compute.  The hmatrix_* buffers are still registered (state_dict contract).

Inside mFormerV1.forward the head GEMMs run in the native plan; the modules' own forward()
(used e.g. by GradNorm-style callers on `feats`) goes through the same HIP GEMM kernels.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import LnxError
from .registry import create_head, register_head


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on lnx_gemm_nt / lnx_gemm_tn (fp32 storage: exact-parity mode)."""

    @staticmethod
    def forward(ctx, x, w, b):
        if not x.is_cuda:
            raise LnxError("linnaeus_amd heads run on the HIP kernels only; move the model and inputs to the GPU")
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        K = x2.shape[1]
        Kp = (K + 3) // 4 * 4
        if Kp != K:
            x2 = torch.nn.functional.pad(x2, (0, Kp - K))
        wp = w.float().contiguous() if Kp == K else torch.nn.functional.pad(w.float(), (0, Kp - K))
        out = torch.empty(x2.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
        ops.gemm_nt(x2, wp, out, bias=b.float() if b is not None else None, dtype=ops.F32)
        ctx.save_for_backward(x2, wp)
        ctx.has_bias = b is not None
        ctx.k = K
        ctx.xshape = x.shape
        return out.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, wp = ctx.saved_tensors
        N, Kp = wp.shape
        g = gy.reshape(-1, N).float()
        Np = (N + 3) // 4 * 4
        gp = torch.nn.functional.pad(g, (0, Np - N)).contiguous() if Np != N else g.contiguous()
        dW = torch.zeros(N, Kp, device=g.device)
        db = torch.zeros(N, device=g.device) if ctx.has_bias else None
        ops.gemm_tn(gp, x2, dW, N=N, db=db, dtype=ops.F32)
        wt = torch.zeros(Kp, Np, device=g.device)
        wt[:, :N] = wp.t()
        dx = torch.empty(x2.shape[0], Kp, device=g.device)
        ops.gemm_nt(gp, wt, dx, dtype=ops.F32)
        return dx[:, : ctx.k].reshape(ctx.xshape), dW[:, : ctx.k], db


def hip_linear(x: torch.Tensor, lin: nn.Linear) -> torch.Tensor:
    return _LinearFn.apply(x, lin.weight, lin.bias)


@register_head("Linear")
class LinearHead(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.fc = nn.Linear(in_features, out_features, bias=bias)

    @property
    def effective_linear(self) -> nn.Linear:
        return self.fc

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return hip_linear(x, self.fc)


class BaseHierarchicalHead(nn.Module):
    """GradNorm mode switch (heads/base_hierarchical_head.py:4-18)."""

    def __init__(self):
        super().__init__()
        self._gradnorm_mode = False

    def set_gradnorm_mode(self, mode: bool) -> None:
        self._gradnorm_mode = bool(mode)

    def is_gradnorm_mode(self) -> bool:
        return self._gradnorm_mode


class _HierarchicalBase(BaseHierarchicalHead):
    def __init__(self, in_features: int, task_key: str, task_keys: List[str], taxonomy_tree: Any, num_classes: Dict[str, int],
                 use_bias: bool = True, level_classifiers_override: Optional[nn.ModuleDict] = None):
        super().__init__()
        if taxonomy_tree is None or not hasattr(taxonomy_tree, "build_hierarchy_matrices"):
            raise TypeError(f"Invalid taxonomy_tree provided to {type(self).__name__}.")
        if task_key not in task_keys:
            raise ValueError(f"Primary task key '{task_key}' not found in task_keys list.")
        if task_key not in num_classes:
            raise ValueError(f"num_classes missing for primary task key '{task_key}'")
        self.in_features = in_features
        self.primary_task_key = task_key
        self.task_keys = task_keys
        self.num_classes = num_classes
        self.taxonomy_tree = taxonomy_tree
        if level_classifiers_override is not None:
            self.level_classifiers = level_classifiers_override
        else:
            self.level_classifiers = nn.ModuleDict()
            for tk in task_keys:
                if num_classes.get(tk) is None:
                    raise ValueError(f"num_classes missing for task '{tk}'")
                self.level_classifiers[tk] = nn.Linear(in_features, num_classes[tk], bias=use_bias)
        self._matrix_keys = []
        for pair_key, matrix in taxonomy_tree.build_hierarchy_matrices().items():
            self.register_buffer(f"hmatrix_{pair_key}", matrix)
            self._matrix_keys.append(pair_key)

    @property
    def effective_linear(self) -> nn.Linear:
        return self.level_classifiers[self.primary_task_key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return hip_linear(x, self.effective_linear)


@register_head("ConditionalClassifier")
class ConditionalClassifierHead(_HierarchicalBase):
    def __init__(self, in_features, task_key, task_keys, taxonomy_tree, num_classes, routing_strategy: str = "soft", temperature: float = 1.0,
                 use_bias: bool = True, level_classifiers_override=None):
        if routing_strategy not in ("soft", "hard", "gumbel"):
            raise ValueError("routing_strategy must be one of ['soft', 'hard', 'gumbel']")
        if temperature <= 0:
            raise ValueError("temperature must be positive.")
        super().__init__(in_features, task_key, task_keys, taxonomy_tree, num_classes, use_bias, level_classifiers_override)
        self.routing_strategy = routing_strategy
        self.temperature = temperature


@register_head("HierarchicalSoftmax")
class HierarchicalSoftmaxHead(_HierarchicalBase):
    def __init__(self, in_features, task_key, task_keys, taxonomy_tree, num_classes, use_bias: bool = True, level_classifiers_override=None):
        super().__init__(in_features, task_key, task_keys, taxonomy_tree, num_classes, use_bias, level_classifiers_override)


def configure_classification_heads(heads_config, in_features: int, num_classes_dict: Optional[Dict[str, int]] = None,
                                   task_keys: Optional[List[str]] = None, taxonomy_tree: Any = None, use_bias: bool = True) -> nn.ModuleDict:
    """Instantiate one head per task of MODEL.CLASSIFICATION.HEADS (heads/utils.py:162-364):
    hierarchical heads share one ModuleDict of per-level Linear classifiers."""
    heads = nn.ModuleDict()
    if not isinstance(heads_config, dict):
        return heads
    hier_types = ("HierarchicalSoftmax", "ConditionalClassifier")
    wants_hier = any(isinstance(c, dict) and str(c.get("TYPE", "")).startswith(hier_types) for c in heads_config.values())
    shared = None
    if wants_hier and task_keys and num_classes_dict:
        shared = nn.ModuleDict()
        for tk in task_keys:
            if num_classes_dict.get(tk) is None:
                raise ValueError(f"num_classes missing for task '{tk}'")
            shared[tk] = nn.Linear(in_features, num_classes_dict[tk], bias=use_bias)
    for task, hc in heads_config.items():
        if not isinstance(hc, dict):
            continue
        ncls = num_classes_dict.get(task) if num_classes_dict else None
        if ncls is None:
            ncls = hc.get("OUT_FEATURES")
            if ncls is None:
                continue
        htype = hc.get("TYPE", "Linear")
        bias = hc.get("USE_BIAS", hc.get("use_bias", use_bias))
        if htype not in hier_types:
            extra = {k: v for k, v in hc.items() if k not in ("TYPE", "IN_FEATURES", "OUT_FEATURES", "USE_BIAS", "use_bias")}
            heads[task] = create_head(htype, in_features=in_features, out_features=ncls, bias=bias, **extra)
            continue
        if not all([task_keys, taxonomy_tree is not None, num_classes_dict, shared is not None]):
            raise ValueError(f"Hierarchical context missing for hierarchical head '{task}'.")
        kw = dict(in_features=in_features, task_key=task, task_keys=task_keys, taxonomy_tree=taxonomy_tree, num_classes=num_classes_dict,
                  use_bias=bias, level_classifiers_override=shared)
        if htype == "ConditionalClassifier":
            kw["routing_strategy"] = hc.get("ROUTING_STRATEGY", hc.get("routing_strategy", "soft"))
            kw["temperature"] = hc.get("TEMPERATURE", hc.get("temperature", 1.0))
        heads[task] = create_head(htype, **kw)
    return heads


def refine_logits_top_down(base: Dict[str, torch.Tensor], heads: nn.ModuleDict, task_keys: List[str]) -> Dict[str, torch.Tensor]:
    """OPT-IN "intended" hierarchical refinement: what heads/conditional_classifier_head.py:176-204
    was written to do and never does (finding F3: the matrix names never match; F4: it would raise if they did).
    Coarsest rank first, every finer rank adds the log prior its parent's routing probabilities induce:

        refined[child] = base[child] + log( softmax(refined[parent] / T) . M[parent -> child] + 1e-10 )

    with M = the tree's hmatrix_{parent}_{child} ([n_parent, n_child] 0/1 membership, utils/taxonomy/taxonomy_tree.py:384).
    Off by default: with it on, outputs intentionally differ from the reference.  The product is lnx_gemm_nt (fp32)."""
    refined = dict(base)
    for i in range(len(task_keys) - 2, -1, -1):
        child, parent = task_keys[i], task_keys[i + 1]
        head = heads[child] if child in heads else None
        name = f"hmatrix_{parent}_{child}"
        if head is None or not hasattr(head, name) or parent not in refined:
            continue
        temp = float(getattr(head, "temperature", 1.0))
        probs = torch.softmax(refined[parent].float() / temp, dim=1)
        prior = _LinearFn.apply(probs, getattr(head, name).t().contiguous().float(), None)  # probs . M
        refined[child] = base[child] + torch.log(prior + 1e-10).to(base[child].dtype)
    return refined
