"""GPU-side image augmentations of the input pipeline: the reference's class names, constructor
arguments and random decisions, with the tensor work done by HIP launches (csrc/aug.hip) on whole batches.

  GPURandomErasing        <- linnaeus/aug/gpu/random_erasing.py:14-94
  GPUAutoAugmentBatch     <- linnaeus/aug/gpu/autoaug.py:13-168   (policies: linnaeus/aug/policies.py)
  GPUAugmentationPipeline <- linnaeus/aug/gpu/pipeline.py:15-103

What the reference does and what is different here
* Random erasing: area / aspect / position / value draws per sample as in the reference (:46-81), but no `.item()`, no Python
  loop over samples: rectangles stay on the device and one launch fills them.  The reference draws positions with
  `torch.randint(0, width - w[valid_idx], ...)`, a tensor as `high`, which raises for more than one valid sample (it only
  runs inside its per-sample pipeline, B = 1); here every sample gets its own uniform position in [0, width - w_i).
* AutoAugment: the sub-policy / operation coin flips are `torch.rand(1).item()` on the CPU generator in the reference's order
  (:92-101), so a seeded run applies the same operations.  Posterize*, Solarize, SolarizeAdd and Invert reproduce the
  reference bit for bit.  The remaining operations name torchvision functions through `torch.nn.functional` (`F.affine`,
  `F.rotate`, `F.adjust_*`, `F.gaussian_blur`: attribute errors at call time) or have one-argument signatures called with two
  (`_auto_contrast`, `_equalize`: type errors), i.e. they cannot run upstream; they are implemented as what they name --
  torchvision's tensor semantics (nearest-neighbour affine with zero fill about the image centre, blend-with-grey saturation,
  blend-with-mean contrast, 3x3 sharpness stencil, reflect-padded Gaussian blur) and the reference's own AutoContrast /
  Equalize formulas -- and tested against torch restatements of those definitions (oracle/aug_oracle.py).
There is no CPU path: images must be on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _lib as L

# sub-policies "Op:prob:magnitude Op:prob:magnitude | ...": the published AutoAugment / EfficientNet tables the reference
# keeps in linnaeus/aug/policies.py (contract data: names, probabilities and magnitudes must be the same to be a drop-in)
_ORIG = ("P:.4:8 Rotate:.6:9|Solarize:.6:5 AutoContrast:.6:5|Equalize:.8:8 Equalize:.6:3|P:.6:7 P:.6:6|Equalize:.4:7 Solarize:.2:4|"
         "Equalize:.4:4 Rotate:.8:8|Solarize:.6:3 Equalize:.6:7|P:.8:5 Equalize:1:2|Rotate:.2:3 Solarize:.6:8|Equalize:.6:8 P:.4:6|"
         "Rotate:.8:8 Color:.4:0|Rotate:.4:9 Equalize:.6:2|Equalize:0:7 Equalize:.8:8|Invert:.6:4 Equalize:1:8|Color:.6:4 Contrast:1:8|"
         "Rotate:.8:8 Color:1:2|Color:.8:8 Solarize:.8:7|Sharpness:.4:7 Invert:.6:8|ShearX:.6:5 Equalize:1:9|Color:.4:0 Equalize:.6:3|"
         "Equalize:.4:7 Solarize:.2:4|Solarize:.6:5 AutoContrast:.6:5|Invert:.6:4 Equalize:1:8|Color:.6:4 Contrast:1:8|Equalize:.8:8 Equalize:.6:3")
_V0R = ("Equalize:.8:1 ShearY:.8:4|Color:.4:9 Equalize:.6:3|Color:.4:1 Rotate:.6:8|Solarize:.8:3 Equalize:.4:7|Solarize:.4:2 Solarize:.6:2|"
        "Color:.2:0 Equalize:.8:8|Equalize:.4:8 SolarizeAdd:.8:3|ShearX:.2:9 Rotate:.6:8|Color:.6:1 Equalize:1:2|Invert:.4:9 Rotate:.6:0|"
        "Equalize:1:9 ShearY:.6:3|Color:.4:7 Equalize:.6:0|PosterizeIncreasing:.4:6 AutoContrast:.4:7|Solarize:.6:8 Color:.6:9|"
        "Solarize:.2:4 Rotate:.8:9|Rotate:1:7 TranslateYRel:.8:9|ShearX:0:0 Solarize:.8:4|ShearY:.8:0 Color:.6:4|Color:1:0 Rotate:.6:2|"
        "Equalize:.8:4 Equalize:0:8|Equalize:1:4 AutoContrast:.6:2|ShearY:.4:7 SolarizeAdd:.6:7|PosterizeIncreasing:.8:2 Solarize:.6:10|"
        "Solarize:.6:8 Equalize:.6:1|Color:.8:6 Rotate:.4:5")
_3A = "Solarize:1:5|Desaturate:1:10|GaussianBlurRand:1:10"


def _parse(spec: str, posterize: str = "PosterizeOriginal") -> List[List[Tuple[str, float, int]]]:
    out = []
    for sub in spec.split("|"):
        ops = []
        for item in sub.split():
            name, p, m = item.split(":")
            ops.append((posterize if name == "P" else name, float(p), int(m)))
        out.append(ops)
    return out


def get_policy(name: str, hparams: Dict[str, Any]) -> List[List[Tuple[str, float, int]]]:
    """linnaeus/aug/policies.py:9-36: 'original', 'originalr' (research posterize), 'v0r', '3a', 'hybrid_v0'; ValueError otherwise."""
    if name == "original":
        return _parse(_ORIG)
    if name == "originalr":
        return _parse(_ORIG, "PosterizeIncreasing")
    if name == "v0r":
        return _parse(_V0R)
    if name == "3a":
        return _parse(_3A)
    if name == "hybrid_v0":
        return _parse(_3A) + _parse(_V0R)
    raise ValueError(f"Unknown AutoAugment policy: {name}")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(images: torch.Tensor, who: str) -> None:
    if not images.is_cuda:
        raise L.LnxError(f"{who} (linnaeus_amd) runs on the MI355X HIP kernels only: images must be on the GPU; there is no CPU fallback")


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def inverse_affine_matrix(angle: float, translate, shear) -> List[float]:
    """Output -> source matrix (2 x 3, coordinates relative to the image centre) of torchvision's `affine` on tensors for
    angle / shear in degrees, translate in pixels, scale 1 (the only forms autoaug.py:52-76 uses)."""
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


class GPUAutoAugmentBatch:
    """`GPUAutoAugmentBatch(policy, color_jitter, config=None)(images [B, C, H, W]) -> images`, float32 in [0, 1]."""

    def __init__(self, policy: str, color_jitter: float, config=None):
        self.hparams = {"color_jitter": color_jitter}
        self.policy = get_policy(policy, self.hparams)
        self._taps: Dict[Any, torch.Tensor] = {}

    # ---- operations (magnitude already scaled by 0.1, autoaug.py:166) ----
    def _point(self, x, op, p0=0.0, p1=0.0, per_img=None):
        per = x[0].numel()
        L.check(L.lib().lnx_aug_pointwise(_ptr(x), C.c_int64(x.numel()), C.c_int64(per), op, C.c_float(p0), C.c_float(p1),
                                          _ptr(per_img) if per_img is not None else None, _stream()), "lnx_aug_pointwise")
        return x

    def _affine(self, x, angle=0.0, translate=(0.0, 0.0), shear=(0.0, 0.0)):
        B, Cn, H, W = x.shape
        y = torch.empty_like(x)
        m = (C.c_float * 6)(*inverse_affine_matrix(angle, translate, shear))
        L.check(L.lib().lnx_aug_affine(_ptr(x), _ptr(y), B * Cn, H, W, m, _stream()), "lnx_aug_affine")
        return y

    def _saturation(self, x, factor):
        B, Cn, H, W = x.shape
        if Cn != 3:
            raise ValueError("Color / Desaturate need three-channel images")
        L.check(L.lib().lnx_aug_saturation(_ptr(x), B, C.c_int64(H * W), C.c_float(factor), _stream()), "lnx_aug_saturation")
        return x

    def _rescale(self, x, rows):
        cols = x.numel() // rows
        mm = torch.empty(2 * rows, device=x.device, dtype=torch.float32)
        L.check(L.lib().lnx_aug_rowstat(_ptr(x), rows, C.c_int64(cols), 0, _ptr(mm), _stream()), "lnx_aug_rowstat")
        L.check(L.lib().lnx_aug_rescale(_ptr(x), C.c_int64(rows), C.c_int64(cols), _ptr(mm), _stream()), "lnx_aug_rescale")
        return x

    def _contrast(self, x, ratio):
        B, Cn, H, W = x.shape
        if Cn != 3 or (Cn * H * W) % 4 != 0:
            raise ValueError("Contrast needs three-channel images whose size is a multiple of 4")
        mean = torch.empty(B, device=x.device, dtype=torch.float32)
        L.check(L.lib().lnx_aug_rowstat(_ptr(x), B, C.c_int64(H * W), 1, _ptr(mean), _stream()), "lnx_aug_rowstat")
        return self._point(x, L.AUG_CONTRAST, ratio, 0.0, mean)

    def _stencil(self, x, taps: List[List[float]], mode: int, ratio: float):
        B, Cn, H, W = x.shape
        k = len(taps)
        key = (x.device, tuple(tuple(r) for r in taps))
        t = self._taps.get(key)
        if t is None:
            t = self._taps[key] = torch.tensor(taps, dtype=torch.float32, device=x.device).contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().lnx_aug_stencil(_ptr(x), _ptr(y), B * Cn, H, W, _ptr(t), k, mode, C.c_float(ratio), _stream()), "lnx_aug_stencil")
        return y

    def _apply_op(self, images: torch.Tensor, op_name: str, magnitude: int) -> torch.Tensor:
        m = magnitude * 0.1
        x = images
        if op_name in ("Posterize", "PosterizeOriginal"):
            return self._point(x, L.AUG_POSTERIZE, 2.0 ** m)
        if op_name == "PosterizeIncreasing":
            return self._point(x, L.AUG_POSTERIZE, 2.0 ** (8 - m))
        if op_name == "Solarize":
            return self._point(x, L.AUG_SOLARIZE, m)
        if op_name == "SolarizeAdd":
            return self._point(x, L.AUG_SOLARIZE_ADD, m, 0.5)
        if op_name == "Invert":
            return self._point(x, L.AUG_INVERT)
        if op_name == "Brightness":
            return self._point(x, L.AUG_BRIGHTNESS, 1.0 + m)
        if op_name == "Contrast":
            return self._contrast(x, 1.0 + m)
        if op_name == "Color":
            return self._saturation(x, 1.0 + m)
        if op_name == "Desaturate":
            return self._saturation(x, 1.0 - m)
        if op_name == "AutoContrast":
            return self._rescale(x, x.shape[0] * x.shape[1])   # per image and channel
        if op_name == "Equalize":
            return self._rescale(x, 1)                          # the reference's simplified form: one range for the tensor
        if op_name == "ShearX":
            return self._affine(x, shear=(m, 0.0))
        if op_name == "ShearY":
            return self._affine(x, shear=(0.0, m))
        if op_name == "TranslateX":
            return self._affine(x, translate=(m, 0.0))
        if op_name == "TranslateY":
            return self._affine(x, translate=(0.0, m))
        if op_name == "TranslateYRel":
            return self._affine(x, translate=(0.0, m * x.shape[-1]))
        if op_name == "Rotate":
            return self._affine(x, angle=-m)  # torchvision's rotate builds the matrix from -angle
        if op_name == "Sharpness":
            return self._stencil(x, [[1 / 13, 1 / 13, 1 / 13], [1 / 13, 5 / 13, 1 / 13], [1 / 13, 1 / 13, 1 / 13]], 1, m)
        if op_name == "GaussianBlurRand":
            k = int(m * 3) * 2 + 1
            if k == 1 or m <= 0.0:
                return x
            half = (k - 1) * 0.5
            g = [math.exp(-0.5 * ((i - half) / m) ** 2) for i in range(k)]
            s = sum(g)
            g = [v / s for v in g]
            return self._stencil(x, [[a * b for b in g] for a in g], 0, 0.0)
        raise ValueError(f"Unknown operation: {op_name}")

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        _need_gpu(images, "GPUAutoAugmentBatch")
        images = images.float().contiguous().clone() if images.dtype != torch.float32 or not images.is_contiguous() else images.clone()
        self._point(images, L.AUG_CLAMP)
        for sub_policy in self.policy:
            if torch.rand(1).item() < self.hparams.get("policy_prob", 1.0):
                for op_name, prob, magnitude in sub_policy:
                    if torch.rand(1).item() < prob:
                        images = self._apply_op(images, op_name, magnitude)  # every kernel ends in clamp(0, 1)
        return images


class GPURandomErasing:
    """`GPURandomErasing(re_config)(images [B, C, H, W]) -> images`; re_config keys PROB, AREA_RANGE, ASPECT_RATIO, COUNT, MODE
    ('const' | 'rand': one uniform value per channel; 'pixel': clamp(N(mean, std) per channel))."""

    def __init__(self, re_config: Dict[str, Any], config=None):
        self.config = re_config
        self._draws: Optional[List[Dict[str, torch.Tensor]]] = None  # tests: the recorded draws of a reference run, one dict per COUNT

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        _need_gpu(images, "GPURandomErasing")
        cfg = self.config
        dev = images.device
        images = images.float().contiguous()
        B, Cn, H, W = images.shape
        inj = self._draws
        gate = (torch.rand(1, device=dev) if inj is None else inj[0]["gate"].to(dev)) <= cfg["PROB"]  # stays on the device: no host sync
        min_area, max_area = cfg["AREA_RANGE"][0] * H * W, cfg["AREA_RANGE"][1] * H * W
        ar = torch.arange(B, device=dev)
        for it in range(cfg["COUNT"]):
            d = inj[it] if inj is not None else None
            areas = torch.empty(B, device=dev).uniform_(min_area, max_area) if d is None else d["areas"].to(dev)
            aspects = torch.empty(B, device=dev).uniform_(*cfg["ASPECT_RATIO"]) if d is None else d["aspects"].to(dev)
            h = torch.sqrt(areas * aspects).round().long()
            w = torch.sqrt(areas / aspects).round().long()
            valid = (w < W) & (h < H) & gate
            if d is None:
                x0 = (torch.rand(B, device=dev) * (W - w).clamp(min=1)).long()
                y0 = (torch.rand(B, device=dev) * (H - h).clamp(min=1)).long()
            else:
                x0, y0 = d["x"].to(dev).long(), d["y"].to(dev).long()
            if cfg["MODE"] in ("const", "rand"):
                vals = torch.empty(B, Cn, device=dev).uniform_(0, 1) if d is None else d["values"].to(dev).reshape(B, Cn)
            else:
                noise = torch.randn(B, Cn, device=dev) if d is None else d["values"].to(dev).reshape(B, Cn)
                vals = (noise * images.std(dim=(2, 3)) + images.mean(dim=(2, 3))).clamp(0, 1)
            zero = torch.zeros_like(h)
            rects = torch.stack([ar, y0, x0, torch.where(valid, h, zero), torch.where(valid, w, zero)], 1).to(torch.int32).contiguous()
            vals = vals.float().contiguous()
            L.check(L.lib().lnx_erase_rects(_ptr(images), B, Cn, H, W, _ptr(rects), _ptr(vals), B, _stream()), "lnx_erase_rects")
        L.check(L.lib().lnx_aug_pointwise(_ptr(images), C.c_int64(images.numel()), C.c_int64(images[0].numel()), L.AUG_CLAMP, C.c_float(0), C.c_float(0), None,
                                          _stream()), "lnx_aug_pointwise")
        return images


class GPUAugmentationPipeline:
    """pipeline.py:15-103: AutoAugment then RandomErasing on `(image, targets, aux_info)`; `image` may be one [C, H, W] sample
    (the reference's per-sample call) or a whole [B, C, H, W] batch.  Values above 1 are taken as 0..255 and divided."""

    def __init__(self, config):
        self.config = config
        self.autoaug = GPUAutoAugmentBatch(config.AUG.AUTOAUG.POLICY, config.AUG.AUTOAUG.COLOR_JITTER, config=config)
        self.random_erasing = GPURandomErasing(config.AUG.RANDOM_ERASE, config=config)

    def __call__(self, sample):
        image, targets, aux_info = sample
        _need_gpu(image, "GPUAugmentationPipeline")
        image = image.float()
        single = image.dim() == 3
        if single:
            image = image.unsqueeze(0)
        scale = torch.where(image.amax() > 1.0, 1.0 / 255.0, 1.0)  # device scalar: no host sync (pipeline.py:74-75)
        image = self.random_erasing(self.autoaug(image * scale))
        return (image.squeeze(0) if single else image), targets, aux_info


def u8hwc_to_f32chw(raw: torch.Tensor) -> torch.Tensor:
    """uint8 [B, H, W, C] (raw stored images, on the GPU) -> float32 [B, C, H, W] in [0, 1]
    (prefetching_h5_dataset.py:214-220's `permute(2, 0, 1).float() / 255.0`, after the transfer instead of before it)."""
    _need_gpu(raw, "u8hwc_to_f32chw")
    if raw.dtype != torch.uint8 or raw.dim() != 4:
        raise ValueError(f"expected a uint8 [B, H, W, C] tensor, got {raw.dtype} {tuple(raw.shape)}")
    raw = raw.contiguous()
    B, H, W, Cn = raw.shape
    out = torch.empty(B, Cn, H, W, device=raw.device, dtype=torch.float32)
    L.check(L.lib().lnx_u8hwc_to_f32chw(_ptr(raw), _ptr(out), B, H, W, Cn, _stream()), "lnx_u8hwc_to_f32chw")
    return out
