// GPU-side batch mixing of the collate step: selective Mixup / CutMix of images and soft targets and the
// chunk-level "hard pick" of metadata, as the reference's GPU augmentations compute them
//   linnaeus/aug/gpu/selective_mixup.py:140-230 (lam x + (1-lam) x[perm]),  :420-560 (metadata chunks)
//   linnaeus/aug/gpu/selective_cutmix.py:200-260 (box paste from x[perm] for samples of a mixable group)
// The reference walks the batch in Python for the metadata (one .item() per sample and chunk); here every piece is one
// launch, HBM-bound: an image batch is read twice (own + partner sample) and written once.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

// out[b, i] = mode 0: lam x[b,i] + (1-lam) x[perm[b],i]
//             mode 1: (valid[b] && i inside the box) ? x[perm[b],i] : x[b,i]      (i = (c, h, w), box over h and w)
//             mode 2: valid[b] ? lam x[b,i] + (1-lam) x[perm[b],i] : x[b,i]
__global__ __launch_bounds__(256) void mix_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ perm, const unsigned char* __restrict__ valid,
                                                       float* __restrict__ out, int B, int64_t row, int H, int W, float lam, int h0, int h1, int w0, int w1,
                                                       int mode) {
    const int64_t nvec = row >> 2;  // row % 4 == 0 (checked by the host)
    const int64_t total = (int64_t)B * nvec;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / nvec);
        const int64_t e = (i % nvec) * 4;
        const int64_t pb = perm[b];
        const float4 a = *reinterpret_cast<const float4*>(x + (int64_t)b * row + e);
        const bool ok = valid == nullptr || valid[b] != 0;
        float4 o = a;
        if (mode == 1) {
            const int w = (int)(e % W), h = (int)((e / W) % H);
            if (ok && h >= h0 && h < h1 && w + 3 >= w0 && w < w1) {
                const float4 p = *reinterpret_cast<const float4*>(x + pb * row + e);
                if (w + 0 >= w0 && w + 0 < w1) o.x = p.x;
                if (w + 1 >= w0 && w + 1 < w1) o.y = p.y;
                if (w + 2 >= w0 && w + 2 < w1) o.z = p.z;
                if (w + 3 >= w0 && w + 3 < w1) o.w = p.w;
            }
        } else if (mode == 0 || ok) {
            const float4 p = *reinterpret_cast<const float4*>(x + pb * row + e);
            const float m = 1.0f - lam;
            o = make_float4(lam * a.x + m * p.x, lam * a.y + m * p.y, lam * a.z + m * p.z, lam * a.w + m * p.w);
        }
        *reinterpret_cast<float4*>(out + (int64_t)b * row + e) = o;
    }
}

// one thread per (sample, chunk): all-or-nothing on both sources (a chunk with any zero entry counts as absent), then
// both present -> own if pick[b] < 0.5 else partner's; one present -> that one; none -> zeros / invalid
__global__ __launch_bounds__(256) void mix_meta_kernel(const float* __restrict__ aux, const unsigned char* __restrict__ mask, const int64_t* __restrict__ perm,
                                                       const float* __restrict__ pick, const int* __restrict__ bounds, int nchunk, int B, int D,
                                                       float* __restrict__ oaux, unsigned char* __restrict__ omask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * nchunk) return;
    const int b = i / nchunk, ch = i % nchunk;
    const int s = bounds[2 * ch], e = bounds[2 * ch + 1];
    const int64_t pb = perm[b];
    bool z1 = false, z2 = false;
    for (int d = s; d < e; ++d) {
        z1 = z1 || aux[(int64_t)b * D + d] == 0.0f;
        z2 = z2 || aux[pb * D + d] == 0.0f;
    }
    int src;  // 0 own, 1 partner, 2 none
    if (!z1 && !z2) src = pick[b] < 0.5f ? 0 : 1;
    else if (!z1) src = 0;
    else if (!z2) src = 1;
    else src = 2;
    const int64_t r = src == 1 ? pb : b;
    for (int d = s; d < e; ++d) {
        oaux[(int64_t)b * D + d] = src == 2 ? 0.0f : aux[r * D + d];
        omask[(int64_t)b * D + d] = src == 2 ? 0 : mask[r * D + d];  // a present chunk has no zero entry: its mask is the source's mask
    }
}

}  // namespace

extern "C" int lnx_mix_rows(const lnx_mix_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->perm && a->out, "lnx_mix_rows: null operand");
    LNX_CHECK(a->B > 0 && a->row > 0 && a->row % 4 == 0, "lnx_mix_rows: row length must be a positive multiple of 4 (got %lld)", (long long)a->row);
    LNX_CHECK(a->mode >= 0 && a->mode <= 2, "lnx_mix_rows: bad mode %d", a->mode);
    if (a->mode == 1) LNX_CHECK(a->H > 0 && a->W > 0 && a->W % 4 == 0 && a->row % ((int64_t)a->H * a->W) == 0, "lnx_mix_rows: box mode needs row = C*H*W with W %% 4 == 0");
    const int64_t total = (int64_t)a->B * (a->row / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(mix_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a->x, a->perm, a->valid, a->out, a->B, a->row, a->H > 0 ? a->H : 1,
                       a->W > 0 ? a->W : (int)a->row, a->lam, a->h0, a->h1, a->w0, a->w1, a->mode);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_mix_meta(const float* aux, const unsigned char* mask, const int64_t* perm, const float* pick, const int* bounds_dev, int nchunk, int B, int D,
                            float* out_aux, unsigned char* out_mask, void* stream) {
    LNX_CHECK(aux && mask && perm && pick && bounds_dev && out_aux && out_mask, "lnx_mix_meta: null operand");
    LNX_CHECK(B > 0 && D > 0 && nchunk > 0, "lnx_mix_meta: bad shape");
    hipLaunchKernelGGL(mix_meta_kernel, dim3(cdiv((int64_t)B * nchunk, 256)), dim3(256), 0, (hipStream_t)stream, aux, mask, perm, pick, bounds_dev, nchunk, B, D,
                       out_aux, out_mask);
    LNX_LAUNCH_CHECK();
    return 0;
}
