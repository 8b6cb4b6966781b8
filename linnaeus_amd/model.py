"""mFormerV1 on the HIP plan: the host-side mirror of the reference's model interface.

Same constructor, `forward(x, meta=None, force_checkpointing=None) -> {task: logits}`,
`forward_features`, `.head` ModuleDict, `parameter_groups_metadata`,
`pretrained_ckpt_handling_metadata` and *identical state_dict names and shapes* as
linnaeus/models/mFormerV1.py:31-541 -- so checkpoints, optimizer parameter filters and
GradNorm code written against the reference keep working -- but no torch.nn forward is
ever executed: the sub-modules below only hold parameters.  forward()/backward() are one
native call each into liblnx_hip.so (lnx_plan_forward / lnx_plan_backward).

There is no CPU path: calling the model with CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from .heads import configure_classification_heads
from .registry import register_model

_DTYPES = {"bf16": L.BF16, "bfloat16": L.BF16, "fp32": L.F32, "float32": L.F32}


class _Cfg(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("batch", C.c_int), ("img_h", C.c_int), ("img_w", C.c_int), ("in_chans", C.c_int),
        ("dims", C.c_int * 4), ("conv_depths", C.c_int * 2), ("rope_depths", C.c_int * 2), ("rope_heads", C.c_int * 2),
        ("mlp_hidden", C.c_int * 2), ("n_meta", C.c_int), ("meta_dims", C.c_int * 8), ("only_last_cls", C.c_int),
        ("n_tasks", C.c_int), ("task_classes", C.c_int * 16), ("inference", C.c_int), ("recompute", C.c_int), ("fp8", C.c_int),
    ]


def _trunc_normal_(t: torch.Tensor, std: float = 0.02) -> torch.Tensor:
    # reference: models/utils/initialization.py:11-36 (a=-2, b=2 absolute bounds)
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


# ---- parameter holders (never called; names/shapes = the reference's state_dict) -------
class _LNCF(nn.Module):  # LayerNormChannelsFirst, blocks/convnext.py:21-30
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Down(nn.Module):  # ConvNeXtDownsampleLayer, blocks/convnext.py:104-110
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm = _LNCF(cin)
        self.conv = nn.Conv2d(cin, cout, kernel_size=2, stride=2)


class _ConvBlk(nn.Module):  # ConvNeXtBlock, blocks/convnext.py:54-71
    def __init__(self, dim: int, ls_init: float, drop_path: float):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        if ls_init <= 0:
            raise NotImplementedError("LAYER_SCALE_INIT_VALUE <= 0 (no LayerScale) is not supported by the HIP path")
        self.gamma = nn.Parameter(ls_init * torch.ones(dim))
        self.drop_prob = float(drop_path)


class _Attn(nn.Module):  # RoPE2DAttention, blocks/rope_2d_mhsa.py:292-303
    def __init__(self, dim: int, heads: int, theta: float):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)
        d = dim // heads
        # learnable mixed frequencies: per head a random direction (rope_2d_mhsa.py:76-111)
        inv = 1.0 / (theta ** (torch.arange(0, d, 2)[: d // 2].float() / d))
        fx, fy = [], []
        for _ in range(heads):
            ang = torch.rand(1) * 2 * math.pi
            fx.append(inv * torch.cos(ang))
            fy.append(inv * torch.sin(ang))
        self.freqs = nn.Parameter(torch.stack([torch.stack(fx, 0), torch.stack(fy, 0)], 0).float())


class _Mlp(nn.Module):  # blocks/mlp.py:35-39
    def __init__(self, cin: int, hidden: int, cout: int):
        super().__init__()
        self.fc1 = nn.Linear(cin, hidden)
        self.fc2 = nn.Linear(hidden, cout)


class _RopeBlk(nn.Module):  # RoPE2DMHSABlock, blocks/rope_2d_mhsa.py:546-574
    def __init__(self, dim: int, heads: int, mlp_ratio: float, theta: float, drop_path: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn = _Attn(dim, heads, theta)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), dim)
        self.drop_prob = float(drop_path)


class _ResNorm(nn.Module):  # ResNormLayer, normalization/res_norm_layer.py:14-21
    def __init__(self, dim: int):
        super().__init__()
        self.norm_fn1 = nn.LayerNorm(dim)
        self.norm_fn2 = nn.LayerNorm(dim)
        self.w1 = nn.Linear(dim, dim)
        self.w2 = nn.Linear(dim, dim)


class _Holder(nn.Sequential):
    """nn.Sequential used purely as an indexed container (keeps '0.', '2.', '3.' names)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the computation runs in the native HIP plan")


class _PlanFn(torch.autograd.Function):
    """Outputs: feats and one [B, classes] logits view PER TASK (views of one buffer made here, inside the Function), so that
    autograd hands the per-task gradients straight to backward() -- slicing the buffer outside would cost a zero-fill, a
    copy and an accumulation per task in the launch-bound gap between forward and backward."""

    @staticmethod
    def forward(ctx, model, x, meta, drop, *params):
        feats, logits = model._plan_forward(x, meta, drop)
        ctx.model = model
        ctx.st = model._active
        ctx.fwd_id = ctx.st["fwd_id"]
        ctx.set_materialize_grads(False)  # an unused output (feats under a logits-only loss) arrives as None, not as zeros
        return (feats, *model._task_views(ctx.st, logits, x.shape[0]))

    @staticmethod
    def backward(ctx, dfeats, *dtasks):
        st = ctx.st
        if st.get("destroyed"):
            raise L.LnxError("backward() of a forward whose plan was evicted from the plan cache (more than "
                             "`max_cached_plans` distinct batch shapes were used in between); raise model.max_cached_plans")
        if st["fwd_id"] != ctx.fwd_id:
            raise L.LnxError("backward() of a forward whose saved activations were overwritten by a later forward of the same "
                             "(batch, image-size) plan; run backward before the next forward")
        if dfeats is None and all(d is None for d in dtasks):
            raise L.LnxError("backward() reached the model with no gradient for any of its outputs")
        dlogits = None
        if st["logits_numel"] > 0:
            # persistent staging buffer in the plan's padded layout; the padding columns are zero from the start and stay so.
            # A task the loss did not use gets a zero gradient (the plan differentiates all heads in one pass).
            dlogits = st.get("dl_buf")
            if dlogits is None:
                dlogits = st["dl_buf"] = torch.zeros(max(st["logits_numel"], 1), device=st["ws"].device, dtype=torch.float32)
                st["dl_views"] = ctx.model._task_views(st, dlogits, st["B"])
            for v, d in zip(st["dl_views"], dtasks):
                if d is None:
                    v.zero_()
                else:
                    v.copy_(d)
        grads = ctx.model._plan_backward(st, dfeats, dlogits)
        return (None, None, None, None, *grads)


@register_model("mFormerV1")
class mFormerV1(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        M = config.MODEL
        # BaseModel._init_common_parameters (models/base_model.py:64-96)
        self.drop_rate = M.DROP_RATE
        self.drop_path_rate = M.DROP_PATH_RATE
        self.attn_drop_rate = M.get("ATTN_DROP_RATE", 0.0)
        self.label_smoothing = M.LABEL_SMOOTHING
        self.only_last_cls = M.ONLY_LAST_CLS
        # Dropout is the identity in eval mode.  In training DROP_RATE (the two Mlp dropouts and proj_drop of every RoPE block,
        # blocks/mlp.py:61-66, rope_2d_mhsa.py:503) and ATTN_DROP_RATE (the attention probabilities, rope_2d_mhsa.py:497) are
        # applied with keep masks drawn per forward (_set_dropout).

        img = M.IMG_SIZE
        self.img_size = (img, img) if isinstance(img, int) else tuple(img)
        in_chans = M.IN_CHANS
        if not hasattr(M, "CONVNEXT_STAGES") and "CONVNEXT_STAGES" not in M:
            raise ValueError("mFormerV1 requires MODEL.CONVNEXT_STAGES config")
        cs = M.CONVNEXT_STAGES
        depths, dims = list(cs.DEPTHS), list(cs.DIMS)
        self.convnext_ls_init = cs.get("LAYER_SCALE_INIT_VALUE", 1e-6)
        if len(depths) != 4 or len(dims) != 4:
            raise ValueError("CONVNEXT_STAGES depths and dims must be lists of length 4.")
        if "ROPE_STAGES" not in M:
            raise ValueError("mFormerV1 requires MODEL.ROPE_STAGES config")
        rs = M.ROPE_STAGES
        rdepths, rdims, rheads, rratio = list(rs.DEPTHS), list(rs.DIMS), list(rs.NUM_HEADS), list(rs.MLP_RATIO)
        self.rope_theta = rs.get("ROPE_THETA", 10000.0)
        self.rope_mixed = rs.get("ROPE_MIXED", True)
        if len(rdepths) != 2 or len(rdims) != 2 or len(rheads) != 2 or len(rratio) != 2:
            raise ValueError("ROPE_STAGES depths, dims, num_heads, mlp_ratio must be lists of length 2.")
        if not self.rope_mixed:
            raise NotImplementedError("ROPE_MIXED=False (axial) is broken in the reference (SURVEY F8) and unused by every config")
        if rdims[0] != dims[2]:
            raise ValueError(f"ConvNeXt dim[2] ({dims[2]}) must match RoPE dim[0] ({rdims[0]})")
        if rdims[1] != dims[3]:
            raise ValueError(f"ConvNeXt dim[3] ({dims[3]}) must match RoPE dim[1] ({rdims[1]})")
        for d_, h_ in zip(rdims, rheads):
            if d_ % h_ != 0 or d_ // h_ != 64:
                raise NotImplementedError(f"HIP attention kernels are built for head_dim 64 (got dim {d_}, heads {h_})")
        self.use_flash_attn = M.get("USE_FLASH_ATTN", False)  # accepted and ignored: attention is always the fused HIP kernel

        # metadata components in IDX order (mFormerV1.py:94-130)
        self.use_meta = False
        self.meta_components: Dict[str, Dict[str, int]] = {}
        self.meta_dims: List[int] = []
        meta_cfg = config.DATA.get("META", None) if hasattr(config, "DATA") else None
        if meta_cfg is not None and meta_cfg.get("ACTIVE", False) and "COMPONENTS" in meta_cfg:
            self.use_meta = True
            items = []
            for name, cc in meta_cfg.COMPONENTS.items():
                if cc.get("ENABLED", False) and cc.get("IDX", -1) >= 0:
                    items.append((cc.get("IDX"), name, cc))
            items.sort(key=lambda t: t[0])
            off = 0
            for _, name, cc in items:
                self.meta_dims.append(cc.DIM)
                self.meta_components[name] = {"dim": cc.DIM, "offset": off}
                off += cc.DIM
        self.extra_token_num = 1 + len(self.meta_dims)

        total = sum(depths[:2]) + sum(rdepths)
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, total)]

        self.stem = _Holder(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4), _LNCF(dims[0]))
        self.downsample_layers = nn.ModuleList([_Down(dims[i], dims[i + 1]) for i in range(3)])
        self.stages = nn.ModuleList()
        k = 0
        for s in range(2):
            self.stages.append(nn.ModuleList([_ConvBlk(dims[s], self.convnext_ls_init, dpr[k + i]) for i in range(depths[s])]))
            k += depths[s]
        for s in range(2):
            self.stages.append(nn.ModuleList([_RopeBlk(rdims[s], rheads[s], rratio[s], self.rope_theta, dpr[k + i]) for i in range(rdepths[s])]))
            k += rdepths[s]
        self.norm_1 = nn.LayerNorm(rdims[0])
        self.norm_2 = nn.LayerNorm(rdims[1])
        self.cls_token_1 = nn.Parameter(torch.zeros(1, 1, rdims[0]))
        self.cls_token_2 = nn.Parameter(torch.zeros(1, 1, rdims[1]))
        _trunc_normal_(self.cls_token_1, std=0.02)
        _trunc_normal_(self.cls_token_2, std=0.02)
        for name, info in self.meta_components.items():
            for s in range(2):
                setattr(self, f"meta_{name.lower()}_head_{s + 1}",
                        _Holder(nn.Linear(info["dim"], rdims[s]), nn.ReLU(inplace=True), nn.LayerNorm(rdims[s]), _ResNorm(rdims[s])))
        if not self.only_last_cls:
            self.cl_1_fc = _Holder(_Mlp(rdims[0], rdims[0], rdims[1]), nn.LayerNorm(rdims[1]))
            self.aggregate = nn.Conv1d(in_channels=2, out_channels=1, kernel_size=1)
        else:
            self.cl_1_fc = None
            self.aggregate = None
        self.final_norm = nn.LayerNorm(rdims[1])

        self.task_keys = list(config.DATA.TASK_KEYS_H5)
        self.head = configure_classification_heads(
            heads_config=M.CLASSIFICATION.HEADS, in_features=rdims[1], num_classes_dict=kwargs.get("num_classes"),
            task_keys=self.task_keys, taxonomy_tree=kwargs.get("taxonomy_tree"))
        self.apply(self._init_weights)

        self._dims, self._depths, self._rdepths, self._rheads = dims, depths, rdepths, rheads
        self._hidden = [int(rdims[s] * rratio[s]) for s in range(2)]
        self._in_chans = in_chans
        dt_name = str(kwargs.get("compute_dtype", os.environ.get("LNX_DTYPE", M.get("LNX_DTYPE", "bf16")))).lower()
        self._fp8 = dt_name in ("fp8", "mxfp8")
        self._dtype_code = L.BF16 if self._fp8 else _DTYPES[dt_name]
        # "autograd" (default): the autograd Function RETURNS the parameter gradients, so AccumulateGrad runs and everything
        # hooked to it works -- torch DistributedDataParallel (what the reference's launch path wraps the model in,
        # main.py:982), torch.autograd.grad() (GradNorm), parameter hooks.  "direct" (opt-in: linnaeus_amd.ddp.DataParallel,
        # bench.py, FusedAdamW loops): gradients are written straight into .grad views of one flat fp32 arena, zero-copy, and
        # autograd sees None for every parameter -- no hook fires, so it must never be combined with torch DDP.
        self.grad_mode = str(kwargs.get("grad_mode", "autograd"))
        self.max_cached_plans = 4     # LRU bound (train, validation, and a tail batch of each) of native plans (each owns a workspace proportional to the batch size)
        self._plans: "OrderedDict[Any, Dict[str, Any]]" = OrderedDict()
        self._active = None
        self._inject_drop = None      # tests: list of per-call [B] multipliers (None entries = no drop)
        self._inject_dropout = None   # tests: uint8 keep-mask buffer in the plan's layout (lnx_plan_set_dropout) instead of a fresh draw
        self._inject_attn_dropout = None  # tests: the same for the attention probabilities (lnx_plan_set_attn_dropout)
        self._segment_hook = None     # DataParallel: called after each backward segment is enqueued
        self._grad_arena = None
        self._grad_views: Optional[List[torch.Tensor]] = None
        self._arena_layout = None
        # set by the reference's train loop / AutoBatch (train.py:93-110, utils/autobatch.py:298-300); together with
        # TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS and forward(force_checkpointing=...) it selects a recompute plan
        self.use_checkpoint = False
        # opt-in: the refinement the hierarchical heads were meant to apply (heads.refine_logits_top_down); the reference's
        # effective behaviour -- and the default here -- is the plain shared Linear per task (finding F3)
        self.hierarchical_refinement = bool(M.CLASSIFICATION.get("HIERARCHICAL_REFINEMENT", False)) if hasattr(M, "CLASSIFICATION") else False

    # reference: mFormerV1._init_weights (mFormerV1.py:351-359)
    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            _trunc_normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # identical contract data to mFormerV1.py:361-405
    @property
    def parameter_groups_metadata(self) -> Dict[str, Any]:
        return {
            "stages": {
                "convnext_stages": ["stem.", "stages.0.", "stages.1.", "downsample_layers.0", "downsample_layers.1"],
                "rope_stages": ["stages.2.", "stages.3.", "downsample_layers.2", "downsample_layers.3"],
                "rope_freqs": ["freqs"],
            },
            "heads": {"classification_heads": ["head."], "meta_heads": ["meta_"]},
            "embeddings": ["cls_token"],
            "norm_layers": ["norm", ".bn", "LayerNorm"],
            "aggregation": ["cl_1_fc.", "aggregate.", "final_norm."],
        }

    @property
    def pretrained_ckpt_handling_metadata(self) -> Dict[str, Any]:
        return {
            "drop_buffers": [],
            "drop_params": ["head.", "meta_", "pos_embed", "norm.", "downsample_layers."],
            "interpolate_rel_pos_bias": False,
            "supports_module_prefix": True,
            "strict": False,
        }

    # ------------------------------------------------------------------ plan plumbing
    def set_compute_dtype(self, name: str) -> None:
        """'bf16' (production: bf16 operands, fp32 accumulate/residual), 'fp32' (strict parity) or 'fp8' (bf16 plus MXFP8
        forward products in the RoPE blocks' qkv / fc1 / fc2 layers: `lnx_mformer_cfg.fp8`)."""
        name = name.lower()
        self._fp8 = name in ("fp8", "mxfp8")
        self._dtype_code = L.BF16 if self._fp8 else _DTYPES[name]
        self.release_plans()

    def set_wgrad_stream(self, on: bool) -> None:
        """Backward scheduling of this model's native plans, present and future (include/lnx.h lnx_plan_set_wgrad_stream): True (the
        default) runs the weight-gradient products on a second HIP stream beside the data-gradient chain, False keeps everything on the
        launch stream.  Same gradients either way; which one is faster depends on what else holds the GPU's hardware queues (bench.py
        times both during warm-up and keeps the faster one)."""
        self._wgrad_stream = bool(on)
        for st in self._plans.values():
            if st.get("handle") is not None and L.lib().lnx_plan_set_wgrad_stream(st["handle"], int(self._wgrad_stream)) < 0:
                L.check(1, "lnx_plan_set_wgrad_stream")

    def set_meta_stream(self, mode: int) -> None:
        """Which stream the metadata-head chains run on beside the launch stream (include/lnx.h lnx_plan_set_meta_stream): 0 the launch stream,
        1 a side stream of their own (default), 2 the weight-gradient stream -- one stream fewer, for steps that share the device's
        hardware queues with a collective library's stream (DataParallel asks for 2)."""
        if mode not in (0, 1, 2):
            raise ValueError(f"meta stream mode {mode!r}: 0 (launch stream), 1 (side stream) or 2 (weight-gradient stream)")
        self._meta_stream = int(mode)
        for st in self._plans.values():
            if st.get("handle") is not None and L.lib().lnx_plan_set_meta_stream(st["handle"], self._meta_stream) < 0:
                L.check(1, "lnx_plan_set_meta_stream")

    def _destroy_plan(self, st) -> None:
        if st.get("destroyed"):
            return
        st["destroyed"] = True
        try:
            L.lib().lnx_plan_destroy(st["handle"])  # synchronises the (process-wide) side streams and destroys the plan's events
        finally:
            st["handle"] = None
            st["ws"] = None       # the workspace goes back to the caching allocator
            st["dl_buf"] = st["dl_views"] = None
            st["dropout_masks"] = st["attn_dropout_masks"] = None
            st["saved_inputs"] = None

    def release_plans(self) -> None:
        """Destroy every cached native plan and free its workspace (e.g. before an AutoBatch probe or after validation)."""
        if torch.cuda.is_available() and self._plans:
            torch.cuda.synchronize()
        for st in list(self._plans.values()):
            self._destroy_plan(st)
        self._plans.clear()
        self._active = None

    @property
    def compute_dtype(self) -> str:
        return "fp8" if self._fp8 else ("bf16" if self._dtype_code == L.BF16 else "fp32")

    def _task_list(self) -> List[str]:
        return list(self.head.keys())

    def _param_for(self, plan_name: str) -> torch.Tensor:
        if plan_name.startswith("meta."):
            _, m, rest = plan_name.split(".", 2)
            comp = list(self.meta_components.keys())[int(m)]
            stage, tail = rest.split(".", 1)  # head_1 / head_2
            return self.get_parameter(f"meta_{comp.lower()}_{stage}.{tail}")
        if plan_name.startswith("head."):
            _, t, kind = plan_name.split(".")
            lin = self.head[self._task_list()[int(t)]].effective_linear
            if kind == "bias" and lin.bias is None:
                raise NotImplementedError("heads without bias are not supported by the HIP plan")
            return lin.weight if kind == "weight" else lin.bias
        return self.get_parameter(plan_name)

    def _wants_recompute(self, force_checkpointing: Optional[bool] = None) -> bool:
        """mFormerV1.py:415-422: `force_checkpointing` wins, else TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS (or the
        `use_checkpoint` attribute the reference's train loop toggles); the blocks apply it only in training mode
        (convnext.py:91, rope_2d_mhsa.py:617)."""
        want = bool(force_checkpointing) if force_checkpointing is not None else self._checkpoint_policy()
        return want and self.training

    def _checkpoint_policy(self) -> bool:
        want = bool(self.use_checkpoint)
        try:
            want = want or bool(self.config.TRAIN.GRADIENT_CHECKPOINTING.ENABLED_NORMAL_STEPS)
        except (AttributeError, KeyError):
            pass
        return want

    def _make_cfg(self, B: int, H: int, W: int, train: bool, recompute: bool = False) -> "_Cfg":
        cfg = _Cfg()
        cfg.inference = 0 if train else 1
        cfg.recompute = 1 if (train and recompute) else 0
        cfg.fp8 = 1 if self._fp8 else 0
        cfg.dtype, cfg.batch, cfg.img_h, cfg.img_w, cfg.in_chans = self._dtype_code, B, H, W, self._in_chans
        cfg.dims[:] = self._dims
        cfg.conv_depths[:] = self._depths[:2]
        cfg.rope_depths[:] = self._rdepths
        cfg.rope_heads[:] = self._rheads
        cfg.mlp_hidden[:] = self._hidden
        cfg.n_meta = len(self.meta_dims)
        for i, d in enumerate(self.meta_dims):
            cfg.meta_dims[i] = d
        cfg.only_last_cls = int(bool(self.only_last_cls))
        tasks = self._task_list()
        cfg.n_tasks = len(tasks)
        for i, t in enumerate(tasks):
            cfg.task_classes[i] = self.head[t].effective_linear.out_features
        return cfg

    def workspace_bytes(self, batch: int, img_h: Optional[int] = None, img_w: Optional[int] = None, train: bool = True,
                        recompute: Optional[bool] = None) -> int:
        """Bytes of plan workspace (saved activations + operand arena + scratch) a forward[/backward] of this batch shape
        needs -- computed by the native planner WITHOUT allocating anything, which is what lets AutoBatch size a batch
        for 288 GB analytically instead of by out-of-memory trials (utils/autobatch.py:111-265)."""
        H = img_h or self.img_size[0]
        W = img_w or img_h or self.img_size[1]
        return self.plan_footprint(batch, H, W, train, recompute)["workspace"]

    def plan_footprint(self, batch: int, img_h: Optional[int] = None, img_w: Optional[int] = None, train: bool = True,
                       recompute: Optional[bool] = None) -> Dict[str, int]:
        """Everything a plan of this shape holds on the device, computed by the native planner without allocating: the
        workspace, the per-forward dropout keep masks that live OUTSIDE it (`lnx_plan_dropout_bytes` /
        `lnx_plan_attn_dropout_bytes`; only when MODEL.DROP_RATE / ATTN_DROP_RATE > 0 and in training) and the fp32 logits
        buffer (the forward's output and, in training, the persistent dlogits buffer of the same size)."""
        lib = L.lib()
        for fn in (lib.lnx_plan_workspace_bytes, lib.lnx_plan_dropout_bytes, lib.lnx_plan_attn_dropout_bytes, lib.lnx_plan_logits_numel):
            fn.restype = C.c_int64
        H = img_h or self.img_size[0]
        W = img_w or img_h or self.img_size[1]
        if recompute is None:  # what a training forward would use
            recompute = self._checkpoint_policy()
        cfg = self._make_cfg(int(batch), int(H), int(W), train, bool(recompute))
        handle = C.c_void_p()
        L.check(lib.lnx_plan_create(C.byref(cfg), C.byref(handle)), "lnx_plan_create")
        try:
            drop = int(lib.lnx_plan_dropout_bytes(handle)) if (train and self.drop_rate > 0.0) else 0
            adrop = int(lib.lnx_plan_attn_dropout_bytes(handle)) if (train and self.attn_drop_rate > 0.0) else 0
            logits = int(lib.lnx_plan_logits_numel(handle)) * 4 * (2 if train else 1)
            return {"workspace": int(lib.lnx_plan_workspace_bytes(handle)) + 256, "dropout": drop, "attn_dropout": adrop, "logits": logits}
        finally:
            lib.lnx_plan_destroy(handle)

    def _get_plan(self, B: int, H: int, W: int, train: bool = True, recompute: bool = False) -> Dict[str, Any]:
        """Native plan for one (batch, image size, dtype, train/inference, recompute) combination.  Inference plans (no_grad /
        frozen model) carry no backward scratch and share activation buffers between blocks; recompute plans (gradient
        checkpointing) keep block inputs only.  The cache is a small LRU: an evicted plan is destroyed and its workspace freed."""
        recompute = bool(train and recompute)
        key = (B, H, W, self._dtype_code, self._fp8, bool(train), recompute)
        st = self._plans.get(key)
        lib = L.lib()
        if st is not None:
            self._plans.move_to_end(key)
        if st is None:
            while len(self._plans) >= max(1, int(self.max_cached_plans)):
                _, old = self._plans.popitem(last=False)
                if old is self._active:
                    self._active = None
                torch.cuda.current_stream().synchronize()
                self._destroy_plan(old)
            cfg = self._make_cfg(B, H, W, train, recompute)
            tasks = self._task_list()
            handle = C.c_void_p()
            L.check(lib.lnx_plan_create(C.byref(cfg), C.byref(handle)), "lnx_plan_create")
            if not getattr(self, "_wgrad_stream", True) and lib.lnx_plan_set_wgrad_stream(handle, 0) < 0:
                L.check(1, "lnx_plan_set_wgrad_stream")
            if getattr(self, "_meta_stream", None) is not None and lib.lnx_plan_set_meta_stream(handle, self._meta_stream) < 0:
                L.check(1, "lnx_plan_set_meta_stream")
            lib.lnx_plan_workspace_bytes.restype = C.c_int64
            lib.lnx_plan_param_name.restype = C.c_char_p
            lib.lnx_plan_param_numel.restype = C.c_int64
            lib.lnx_plan_logits_numel.restype = C.c_int64
            lib.lnx_plan_logits_offset.restype = C.c_int64
            n = lib.lnx_plan_num_params(handle)
            names = [lib.lnx_plan_param_name(handle, i).decode() for i in range(n)]
            params = [self._param_for(nm) for nm in names]
            for nm, p_, i in zip(names, params, range(n)):
                if p_.numel() != lib.lnx_plan_param_numel(handle, i):
                    raise L.LnxError(f"parameter {nm}: module has {p_.numel()} elements, plan expects {lib.lnx_plan_param_numel(handle, i)}")
            dev = params[0].device
            wsb = lib.lnx_plan_workspace_bytes(handle)
            ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
            seg_of = [3] * n
            buf = (C.c_int * n)()
            for seg in range(4):
                cnt = lib.lnx_plan_segment_params(handle, seg, buf, n)
                for j in range(cnt):
                    seg_of[buf[j]] = seg
            st = dict(handle=handle, names=names, params=params, ws=ws, ptrs=None, n=n, seg_of=seg_of,
                      ndrop=lib.lnx_plan_num_drop_calls(handle), logits_numel=lib.lnx_plan_logits_numel(handle),
                      logit_off=[lib.lnx_plan_logits_offset(handle, i) for i in range(len(tasks))],
                      logit_ld=[lib.lnx_plan_logits_ld(handle, i) for i in range(len(tasks))], tasks=tasks, B=B, fwd_id=0, train=bool(train))
            self._plans[key] = st
        self._ensure_bound(st)
        return st

    @staticmethod
    def _arena_offsets(numels: List[int], seg_of: List[int]):
        """Layout of the flat gradient arena: parameters ordered by (backward segment, plan index), 16-byte aligned slices.
        Returns (offset per plan index, total floats, {segment: (lo, hi)})."""
        order = sorted(range(len(numels)), key=lambda i: (seg_of[i], i))
        offs, cur, seg_bounds = {}, 0, {}
        for i in order:
            seg = seg_of[i]
            seg_bounds.setdefault(seg, [cur, cur])
            offs[i] = cur
            cur += (numels[i] + 3) // 4 * 4
            seg_bounds[seg][1] = cur
        return offs, cur, {s: tuple(b) for s, b in seg_bounds.items()}

    def grad_arena_layout(self, batch: int = 1, img_h: Optional[int] = None, img_w: Optional[int] = None) -> Dict[str, Any]:
        """Host-only (no GPU, nothing allocated): the gradient-arena geometry the data-parallel reducer works on -- plan
        parameter names, the module tensor behind each (shared hierarchical-head Linears appear ONCE), its backward segment
        (`lnx_plan_segment_params`), its slice of the arena, and the four bucket bounds."""
        lib = L.lib()
        lib.lnx_plan_param_name.restype = C.c_char_p
        lib.lnx_plan_param_numel.restype = C.c_int64
        H = img_h or self.img_size[0]
        W = img_w or img_h or self.img_size[1]
        cfg = self._make_cfg(int(batch), int(H), int(W), True, False)
        handle = C.c_void_p()
        L.check(lib.lnx_plan_create(C.byref(cfg), C.byref(handle)), "lnx_plan_create")
        try:
            n = lib.lnx_plan_num_params(handle)
            names = [lib.lnx_plan_param_name(handle, i).decode() for i in range(n)]
            numels = [int(lib.lnx_plan_param_numel(handle, i)) for i in range(n)]
            seg_of = [3] * n
            buf = (C.c_int * n)()
            for seg in range(4):
                for j in range(lib.lnx_plan_segment_params(handle, seg, buf, n)):
                    seg_of[buf[j]] = seg
        finally:
            lib.lnx_plan_destroy(handle)
        params = [self._param_for(nm) for nm in names]
        offs, total, bounds = self._arena_offsets(numels, seg_of)
        return dict(names=names, params=params, numels=numels, seg_of=seg_of, offsets=[offs[i] for i in range(n)], total=total, bounds=bounds)

    def _ensure_grad_arena(self, st) -> None:
        """One flat fp32 gradient arena, ordered by backward segment (so each segment is one
        contiguous all-reduce bucket); every parameter gets a view into it."""
        params = st["params"]
        ident = tuple(id(p) for p in params)
        dev = params[0].device
        if self._grad_arena is not None and self._arena_layout == (ident, dev):
            return
        offs, cur, seg_bounds = self._arena_offsets([p_.numel() for p_ in params], st["seg_of"])
        self._grad_arena = torch.zeros(cur, dtype=torch.float32, device=dev)
        self._grad_views = [self._grad_arena[offs[i]: offs[i] + params[i].numel()].view(params[i].shape) for i in range(st["n"])]
        self._segment_bounds = seg_bounds
        self._arena_layout = (ident, dev)
        for s2 in self._plans.values():
            s2["ptrs"] = None

    def _ensure_bound(self, st) -> None:
        params = st["params"]
        if not params[0].is_cuda:
            raise L.LnxError("mFormerV1 (linnaeus_amd) runs on the MI355X HIP kernels only: call model.cuda() first; there is no CPU fallback")
        for p_ in params:
            if p_.dtype != torch.float32 or not p_.is_contiguous():
                raise L.LnxError("parameters must be contiguous fp32 (master weights); the bf16 operand copies are made by the plan")
        if st["train"]:
            self._ensure_grad_arena(st)
        ptrs = tuple(p_.data_ptr() for p_ in params)
        if st["ptrs"] == ptrs:
            return
        n = st["n"]
        parr = (C.c_void_p * n)(*ptrs)
        garr = (C.c_void_p * n)(*[v.data_ptr() for v in self._grad_views]) if st["train"] else None
        wsp = (st["ws"].data_ptr() + 255) // 256 * 256
        L.check(L.lib().lnx_plan_bind(st["handle"], parr, garr, C.c_void_p(wsp)), "lnx_plan_bind")
        st["ptrs"] = ptrs

    def _draw_drop_scales(self, st, B: int, dev) -> Optional[torch.Tensor]:
        """Per-call, per-sample DropPath multipliers floor(keep + U)/keep (blocks/drop_path.py:29-33);
        ConvNeXt blocks draw once, RoPE blocks twice (attn and mlp branches)."""
        probs: List[float] = []
        for s in range(2):
            probs += [blk.drop_prob for blk in self.stages[s]]
        for s in range(2, 4):
            for blk in self.stages[s]:
                probs += [blk.drop_prob, blk.drop_prob]
        assert len(probs) == st["ndrop"]
        if self._inject_drop is not None:
            inj = list(self._inject_drop)
            assert len(inj) == len(probs)
            mask = bytes(int(v is not None) for v in inj)
            if not any(mask):
                return None
            rows = [v.to(dev, torch.float32) if v is not None else torch.ones(B, device=dev) for v in inj]
            st["drop_mask"] = mask
            return torch.stack(rows, 0).contiguous()
        if not self.training or all(p == 0.0 for p in probs):
            return None
        cache = st.get("drop_keep")
        if cache is None or cache[0] != tuple(probs) or cache[1].device != dev:
            # built once per plan: a host->device copy every step would stall the launch queue
            keep = torch.tensor([1.0 - p for p in probs], device=dev).unsqueeze(1)
            cache = st["drop_keep"] = (tuple(probs), keep, bytes(int(p > 0.0) for p in probs))
        _, keep, mask = cache
        # floor(keep + U) is Bernoulli(keep): one RNG launch + one divide instead of rand/add/floor/div
        scales = torch.bernoulli(keep.expand(len(probs), B)).div_(keep)
        st["drop_mask"] = mask
        return scales

    def _task_views(self, st, logits: torch.Tensor, B: int) -> List[torch.Tensor]:
        """[B, classes] views, one per task, of a flat buffer in the plan's layout (task t: [B, ld_t] at offset off_t)."""
        out = []
        for i, t in enumerate(st["tasks"]):
            ld = st["logit_ld"][i]
            nc = self.head[t].effective_linear.out_features
            out.append(logits[st["logit_off"][i]: st["logit_off"][i] + B * ld].view(B, ld)[:, :nc])
        return out

    def _set_dropout(self, st, train: bool, dev) -> None:
        """MODEL.DROP_RATE (the two Mlp dropouts and proj_drop of every RoPE block: blocks/mlp.py:61-66, rope_2d_mhsa.py:503) and
        MODEL.ATTN_DROP_RATE (the attention probabilities: rope_2d_mhsa.py:497) in training mode: draw this forward's keep
        masks and hand them to the plan; they stay alive in `st` until the next forward of this plan (its backward reads them)."""
        lib = L.lib()
        active = train and self.training
        for key, rate, inject, nbytes_fn, set_fn in (
                ("dropout_masks", self.drop_rate, self._inject_dropout, lib.lnx_plan_dropout_bytes, lib.lnx_plan_set_dropout),
                ("attn_dropout_masks", self.attn_drop_rate, self._inject_attn_dropout, lib.lnx_plan_attn_dropout_bytes, lib.lnx_plan_set_attn_dropout)):
            if not (active and rate > 0.0):
                if st.get(key) is not None:
                    L.check(set_fn(st["handle"], None, C.c_float(0.0)), "lnx_plan_set_dropout")
                    st[key] = None
                continue
            if key == "dropout_masks" and self._fp8:
                raise NotImplementedError("DROP_RATE > 0 is not available in fp8 mode")
            nbytes_fn.restype = C.c_int64
            nbytes = int(nbytes_fn(st["handle"]))
            if inject is not None:
                masks = inject.to(dev, torch.uint8).contiguous()
                assert masks.numel() == nbytes, (key, masks.numel(), nbytes)
            else:
                masks = torch.empty(nbytes, dtype=torch.uint8, device=dev).bernoulli_(1.0 - rate)
            L.check(set_fn(st["handle"], C.c_void_p(masks.data_ptr()), C.c_float(rate)), "lnx_plan_set_dropout")
            st[key] = masks

    def _plan_forward(self, x, meta, drop):
        st = self._active
        B = x.shape[0]
        feats = torch.empty(B, self._dims[3], device=x.device, dtype=torch.float32)
        logits = torch.empty(max(st["logits_numel"], 1), device=x.device, dtype=torch.float32)  # every exposed element is written by the head GEMMs
        mask = st.get("drop_mask") if drop is not None else None
        L.check(L.lib().lnx_plan_forward(
            st["handle"], C.c_void_p(x.data_ptr()), C.c_void_p(meta.data_ptr()) if meta is not None else None,
            C.c_void_p(drop.data_ptr()) if drop is not None else None, mask, C.c_void_p(feats.data_ptr()),
            C.c_void_p(logits.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lnx_plan_forward")
        st["saved_inputs"] = (x, meta, drop)  # keep alive until backward
        st["fwd_id"] = st.get("fwd_id", 0) + 1
        return feats, logits

    def _plan_backward(self, st, dfeats, dlogits):
        direct = self.grad_mode == "direct"
        params = st["params"]
        views = self._grad_views
        # The kernels ACCUMULATE into the arena.  A slice may keep its contents only if the parameter's .grad IS that
        # slice (gradient accumulation in direct mode); every other slice (grad None, or a foreign tensor such as a clone
        # left by autograd mode / zero_grad(set_to_none=False)) must start from zero.
        if direct:
            alias = [p_.grad is not None and p_.grad.data_ptr() == v.data_ptr() for p_, v in zip(params, views)]
            if not any(alias):
                self._grad_arena.zero_()
            elif not all(alias):
                for a, v in zip(alias, views):
                    if not a:
                        v.zero_()
        else:
            alias = None
            self._grad_arena.zero_()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        dl = dlogits.contiguous() if dlogits is not None else None
        df = dfeats.contiguous() if dfeats is not None else None
        if st["logits_numel"] == 0:
            dl = None
        lib = L.lib()
        if self._segment_hook is None:
            L.check(lib.lnx_plan_backward(st["handle"], C.c_void_p(dl.data_ptr()) if dl is not None else None,
                                          C.c_void_p(df.data_ptr()) if df is not None else None, -1, stream), "lnx_plan_backward")
        else:
            for seg in range(4):
                L.check(lib.lnx_plan_backward(st["handle"], C.c_void_p(dl.data_ptr()) if dl is not None else None,
                                              C.c_void_p(df.data_ptr()) if df is not None else None, seg, stream), "lnx_plan_backward")
                self._segment_hook(seg)
        if direct:
            for p_, v, a in zip(params, views, alias):
                if p_.grad is None:
                    p_.grad = v
                elif not a:
                    p_.grad.add_(v)
            return [None] * len(params)
        # autograd mode: hand out copies (ONE copy of the arena, sliced), so nothing a caller holds ever aliases the arena
        # the next backward zeroes
        snap = self._grad_arena.clone()
        base = self._grad_arena.data_ptr()
        return [snap[(v.data_ptr() - base) // 4: (v.data_ptr() - base) // 4 + v.numel()].view(v.shape) for v in views]

    # ------------------------------------------------------------------ public forward
    def _run(self, x: torch.Tensor, meta: Optional[torch.Tensor], force_checkpointing: Optional[bool] = None):
        if not x.is_cuda:
            raise L.LnxError("mFormerV1 (linnaeus_amd) has no CPU path: inputs must be on the GPU")
        if x.dim() != 4 or x.shape[1] != self._in_chans:
            raise ValueError(f"expected input [B, {self._in_chans}, H, W], got {tuple(x.shape)}")
        x = x.float().contiguous()
        B, _, H, W = x.shape
        if self.use_meta and self.meta_dims:
            if meta is None:
                # the reference asserts N == H*W + extra_token_num inside attention (rope_2d_mhsa.py:427-429)
                raise AssertionError("metadata components are configured but meta is None")
            if meta.shape[-1] != sum(self.meta_dims):
                raise ValueError(f"meta must be [B, {sum(self.meta_dims)}] (components in IDX order), got {tuple(meta.shape)}")
            meta = meta.float().contiguous()
        else:
            meta = None
        train = torch.is_grad_enabled() and any(p_.requires_grad for p_ in self.parameters())
        st = self._get_plan(B, H, W, train, self._wants_recompute(force_checkpointing))
        self._active = st
        drop = self._draw_drop_scales(st, B, x.device)
        self._set_dropout(st, train, x.device)
        if train:
            feats, *views = _PlanFn.apply(self, x, meta, drop, *st["params"])
        else:
            feats, logits = self._plan_forward(x, meta, drop)
            views = self._task_views(st, logits, B)
        return st, feats, views

    def forward_features(self, x: torch.Tensor, meta: Optional[torch.Tensor] = None, force_checkpointing: Optional[bool] = None) -> torch.Tensor:
        """[B, D3] features after final_norm (mFormerV1.py:407-529).  `force_checkpointing` / the config's
        GRADIENT_CHECKPOINTING flag select a recompute plan (block inputs kept, activations recomputed in backward)."""
        return self._run(x, meta, force_checkpointing)[1]

    def forward(self, x: torch.Tensor, meta: Optional[torch.Tensor] = None, force_checkpointing: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        st, feats, views = self._run(x, meta, force_checkpointing)
        acast = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None
        out = {}
        for t, v in zip(st["tasks"], views):
            out[t] = v
            if acast is not None:
                out[t] = out[t].to(acast)  # the reference returns logits in the autocast dtype (SURVEY 8b "Tensor conventions")
        self._last_feats = feats
        if self.hierarchical_refinement:
            from .heads import refine_logits_top_down

            out = refine_logits_top_down(out, self.head, self.task_keys)
        return out

    def __del__(self):
        try:
            for st in self._plans.values():
                self._destroy_plan(st)
        except Exception:
            pass
