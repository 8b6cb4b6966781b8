// GPU-side single-image / batch augmentations of the input pipeline (SURVEY 8f-3 remainder): the tensor operations behind
// the reference's GPURandomErasing and GPUAutoAugmentBatch, and the uint8 NHWC -> fp32 NCHW conversion of a raw image batch
//   linnaeus/aug/gpu/random_erasing.py:24-94   (rectangles filled with one value per channel, result clamped to [0, 1])
//   linnaeus/aug/gpu/autoaug.py:44-168         (one clamp(0, 1) after every operation)
//   linnaeus/h5data/prefetching_h5_dataset.py:214-220 (uint8 HWC image -> float CHW / 255)
// Images are fp32 [B, C, H, W], contiguous.  Every kernel is HBM-bound (one read + one write of the batch, 16-byte accesses
// where the shape allows); the decisions (which operation, which rectangle) are drawn on the host exactly as the reference
// draws them, so a seeded run makes the same choices.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// per-element operations; p0 / p1 are the operation's scalars, `per_img` (optional) one scalar per image
__device__ __forceinline__ float point_op(float v, int op, float p0, float p1, float s) {
    switch (op) {
        case LNX_AUG_POSTERIZE: return floorf(v * 255.0f / p0) * p0 / 255.0f;         // p0 = 2^bits (autoaug.py:117-119)
        case LNX_AUG_SOLARIZE: return v < p0 ? v : 1.0f - v;                           // :130-131
        case LNX_AUG_SOLARIZE_ADD: return v < p1 ? clamp01(v + p0) : v;                // :133-138
        case LNX_AUG_INVERT: return 1.0f - v;                                          // :86
        case LNX_AUG_BRIGHTNESS: return v * p0;                                        // blend with black, ratio p0
        case LNX_AUG_CONTRAST: return p0 * v + (1.0f - p0) * s;                        // blend with the image's mean grey level s
        default: return v;                                                             // LNX_AUG_CLAMP
    }
}

__global__ __launch_bounds__(256) void pointwise_kernel(float* __restrict__ x, int64_t n, int64_t per_image, int op, float p0, float p1,
                                                        const float* __restrict__ per_img) {
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        const float s = per_img ? per_img[(i * 4) / per_image] : 0.0f;  // per_image % 4 == 0 (host check): one image per vector
        v.x = clamp01(point_op(v.x, op, p0, p1, s));
        v.y = clamp01(point_op(v.y, op, p0, p1, s));
        v.z = clamp01(point_op(v.z, op, p0, p1, s));
        v.w = clamp01(point_op(v.w, op, p0, p1, s));
        reinterpret_cast<float4*>(x)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
        x[i] = clamp01(point_op(x[i], op, p0, p1, per_img ? per_img[i / per_image] : 0.0f));
    }
}

// saturation: out_c = clamp(f c + (1 - f) grey), grey = 0.2989 r + 0.587 g + 0.114 b (three-channel images)
__global__ __launch_bounds__(256) void saturation_kernel(float* __restrict__ x, int B, int64_t hw, float f) {
    const int64_t total = (int64_t)B * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        float* r = x + (b * 3) * hw + p;
        const float cr = r[0], cg = r[hw], cb = r[2 * hw];
        const float grey = 0.2989f * cr + 0.587f * cg + 0.114f * cb;
        r[0] = clamp01(f * cr + (1.0f - f) * grey);
        r[hw] = clamp01(f * cg + (1.0f - f) * grey);
        r[2 * hw] = clamp01(f * cb + (1.0f - f) * grey);
    }
}

// row statistics: one workgroup per row; kind 0: min and max -> out[2 row], out[2 row + 1]; kind 1: mean grey level of a
// three-channel image (row = image, cols = H W) -> out[row]
__global__ __launch_bounds__(256) void rowstat_kernel(const float* __restrict__ x, int64_t cols, int kind, float* __restrict__ out) {
    const int row = blockIdx.x;
    float a = kind == 0 ? 3.4e38f : 0.0f, b = -3.4e38f;
    if (kind == 0) {
        const float* r = x + (int64_t)row * cols;
        for (int64_t i = threadIdx.x; i < cols; i += 256) {
            a = fminf(a, r[i]);
            b = fmaxf(b, r[i]);
        }
    } else {
        const float* r = x + (int64_t)row * 3 * cols;
        for (int64_t i = threadIdx.x; i < cols; i += 256) a += 0.2989f * r[i] + 0.587f * r[cols + i] + 0.114f * r[2 * cols + i];
    }
    __shared__ float sa[4], sb[4];
    if (kind == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = fminf(a, __shfl_xor(a, o, 64));
            b = fmaxf(b, __shfl_xor(b, o, 64));
        }
    } else {
        a = wave_sum(a);
    }
    if ((threadIdx.x & 63) == 0) {
        sa[threadIdx.x >> 6] = a;
        sb[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (kind == 0) {
            out[2 * row] = fminf(fminf(sa[0], sa[1]), fminf(sa[2], sa[3]));
            out[2 * row + 1] = fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3]));
        } else {
            out[row] = (sa[0] + sa[1] + sa[2] + sa[3]) / (float)cols;
        }
    }
}

// x[row, :] = clamp((x - min_row) / (max_row - min_row + 1e-6)); the reference's AutoContrast / Equalize forms (autoaug.py:143-151)
__global__ __launch_bounds__(256) void rescale_kernel(float* __restrict__ x, int64_t rows, int64_t cols, const float* __restrict__ mm) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const float lo = mm[2 * r], hi = mm[2 * r + 1];
        x[i] = clamp01((x[i] - lo) / (hi - lo + 1e-6f));
    }
}

// y[b, c, h, w] = x[b, c, round(sy), round(sx)] with (sx, sy) = M (w - cx, h - cy) + (cx, cy); nearest neighbour (round half to
// even, as grid_sample), zero outside: torchvision's tensor affine / rotate with its default interpolation and fill
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W, float m00, float m01, float m02,
                                                     float m10, float m11, float m12) {
    const int64_t total = (int64_t)planes * H * W;
    const float cx = 0.5f * (W - 1), cy = 0.5f * (H - 1);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const int64_t pl = i / ((int64_t)H * W);
        const float dx = w - cx, dy = h - cy;
        const float sx = rintf(m00 * dx + m01 * dy + m02 + cx), sy = rintf(m10 * dx + m11 * dy + m12 + cy);
        float v = 0.0f;
        if (sx >= 0.0f && sx <= (float)(W - 1) && sy >= 0.0f && sy <= (float)(H - 1)) v = x[(pl * H + (int)sy) * W + (int)sx];
        y[i] = clamp01(v);
    }
}

// y = clamp(ratio x + (1 - ratio) (x * taps)), k x k taps (k odd, <= 15); mode 0: reflect padding (Gaussian blur),
// mode 1: border pixels keep x (torchvision's sharpness stencil)
__global__ __launch_bounds__(256) void stencil_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int H, int W, const float* __restrict__ taps, int k,
                                                      int mode, float ratio) {
    const int64_t total = (int64_t)planes * H * W;
    const int r = k >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H);
        const float* p = x + (i / ((int64_t)H * W)) * H * W;
        const float self = p[h * W + w];
        float acc = 0.0f;
        if (mode == 1 && (w < r || h < r || w >= W - r || h >= H - r)) {
            acc = self;
        } else {
            for (int dy = -r; dy <= r; ++dy) {
                int hh = h + dy;
                hh = hh < 0 ? -hh : (hh >= H ? 2 * H - 2 - hh : hh);
                for (int dx = -r; dx <= r; ++dx) {
                    int ww = w + dx;
                    ww = ww < 0 ? -ww : (ww >= W ? 2 * W - 2 - ww : ww);
                    acc += taps[(dy + r) * k + dx + r] * p[hh * W + ww];
                }
            }
        }
        y[i] = clamp01(ratio * self + (1.0f - ratio) * acc);
    }
}

// rects[j] = (image, y0, x0, h, w); vals[j, c]: one workgroup column per rectangle
// (the list lives on the device, the host entry cannot check it: a rectangle of another image, or an empty one, is skipped and
// every pixel is clipped to the image)
__global__ __launch_bounds__(256) void erase_kernel(float* __restrict__ x, int B, int C, int H, int W, const int* __restrict__ rects, const float* __restrict__ vals) {
    const int j = blockIdx.y;
    const int b = rects[5 * j], y0 = rects[5 * j + 1], x0 = rects[5 * j + 2], h = rects[5 * j + 3], w = rects[5 * j + 4];
    if (b < 0 || b >= B || h <= 0 || w <= 0) return;
    const int64_t n = (int64_t)C * h * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int ww = (int)(i % w), hh = (int)((i / w) % h), c = (int)(i / ((int64_t)w * h));
        const int yy = y0 + hh, xx = x0 + ww;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) x[(((int64_t)b * C + c) * H + yy) * W + xx] = vals[(int64_t)j * C + c];
    }
}

// raw image batch: uint8 [B, H, W, C] (as stored) -> fp32 [B, C, H, W] / 255
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t b = i / ((int64_t)W * H * C);
        dst[i] = (float)src[((b * H + h) * W + w) * C + c] / 255.0f;
    }
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int lnx_aug_pointwise(float* x, int64_t n, int64_t per_image, int op, float p0, float p1, const float* per_image_scalar, void* stream) {
    LNX_CHECK(x && n > 0, "lnx_aug_pointwise: empty tensor");
    LNX_CHECK(op >= LNX_AUG_CLAMP && op <= LNX_AUG_CONTRAST, "lnx_aug_pointwise: unknown operation %d", op);
    if (per_image_scalar) LNX_CHECK(per_image > 0 && per_image % 4 == 0, "lnx_aug_pointwise: per-image scalars need an image size that is a multiple of 4");
    if (op == LNX_AUG_POSTERIZE) LNX_CHECK(p0 > 0.0f, "lnx_aug_pointwise: posterize needs 2^bits > 0");
    LNX_CHECK((((uintptr_t)x) & 15) == 0, "lnx_aug_pointwise: tensor must be 16-byte aligned");
    hipLaunchKernelGGL(pointwise_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, n, per_image > 0 ? per_image : n, op, p0, p1, per_image_scalar);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_aug_saturation(float* x, int B, int64_t hw, float factor, void* stream) {
    LNX_CHECK(x && B > 0 && hw > 0, "lnx_aug_saturation: empty tensor");
    hipLaunchKernelGGL(saturation_kernel, dim3(grid_for((int64_t)B * hw)), dim3(256), 0, (hipStream_t)stream, x, B, hw, factor);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_aug_rowstat(const float* x, int rows, int64_t cols, int kind, float* out, void* stream) {
    LNX_CHECK(x && out && rows > 0 && cols > 0, "lnx_aug_rowstat: empty tensor");
    LNX_CHECK(kind == 0 || kind == 1, "lnx_aug_rowstat: kind must be 0 (min / max) or 1 (mean grey level of RGB rows)");
    hipLaunchKernelGGL(rowstat_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, cols, kind, out);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_aug_rescale(float* x, int64_t rows, int64_t cols, const float* minmax, void* stream) {
    LNX_CHECK(x && minmax && rows > 0 && cols > 0, "lnx_aug_rescale: empty tensor");
    hipLaunchKernelGGL(rescale_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, x, rows, cols, minmax);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_aug_affine(const float* x, float* y, int planes, int H, int W, const float* m6, void* stream) {
    LNX_CHECK(x && y && x != y && m6 && planes > 0 && H > 0 && W > 0, "lnx_aug_affine: bad arguments (out of place only)");
    hipLaunchKernelGGL(affine_kernel, dim3(grid_for((int64_t)planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, m6[0], m6[1], m6[2], m6[3], m6[4],
                       m6[5]);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_aug_stencil(const float* x, float* y, int planes, int H, int W, const float* taps_dev, int k, int mode, float ratio, void* stream) {
    LNX_CHECK(x && y && x != y && taps_dev && planes > 0, "lnx_aug_stencil: bad arguments (out of place only)");
    LNX_CHECK(k >= 1 && k <= 15 && (k & 1) == 1 && H > k / 2 && W > k / 2, "lnx_aug_stencil: kernel size %d must be odd, <= 15 and smaller than the image", k);
    LNX_CHECK(mode == 0 || mode == 1, "lnx_aug_stencil: mode must be 0 (reflect) or 1 (keep borders)");
    hipLaunchKernelGGL(stencil_kernel, dim3(grid_for((int64_t)planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, taps_dev, k, mode, ratio);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_erase_rects(float* x, int B, int C, int H, int W, const int* rects_dev, const float* values_dev, int n_rects, void* stream) {
    LNX_CHECK(x && B > 0 && C > 0 && H > 0 && W > 0, "lnx_erase_rects: empty tensor");
    if (n_rects == 0) return 0;
    LNX_CHECK(n_rects > 0 && n_rects <= 65535 && rects_dev && values_dev, "lnx_erase_rects: bad rectangle list (%d)", n_rects);
    hipLaunchKernelGGL(erase_kernel, dim3(16, n_rects), dim3(256), 0, (hipStream_t)stream, x, B, C, H, W, rects_dev, values_dev);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_u8hwc_to_f32chw(const unsigned char* src, float* dst, int B, int H, int W, int C, void* stream) {
    LNX_CHECK(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "lnx_u8hwc_to_f32chw: empty tensor");
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, B, H, W, C);
    LNX_LAUNCH_CHECK();
    return 0;
}
