// Depthwise 7x7 convolution (padding 3, NHWC) for gfx950: forward / data-gradient and
// weight-gradient kernels.  Reference: nn.Conv2d(dim, dim, 7, padding=3, groups=dim),
// blocks/convnext.py:56-58.
//
// HBM-bound op: algorithmic traffic is one read of x and one write of y per element
// (49 MACs per element).  To get there the input tile with its 3-pixel halo is staged once
// in LDS (channel-innermost, so a wave reads 32 consecutive channels = conflict-free
// ds_read_b32) and every thread produces a 16-pixel output row segment from registers:
// 22 LDS reads + 7 weight loads per 112 FMAs.
//
// Tile: 8 rows x 16 cols x 32 channels of output per 256-thread workgroup
// (thread = (channel lane 0..31, row 0..7)); LDS 14 x 22 x 32 fp32 = 38.5 KiB.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int TH = 8, TW = 16, CB = 32;
constexpr int IH = TH + 6, IW = TW + 6;

struct DwP {
    const void* x;
    const float* w49;
    const float* bias;
    const float* res;
    void* y;
    int B, H, W, C;
    int flip;
    int tiles_h, tiles_w, cblocks;
};

template <typename TX>
__device__ __forceinline__ void load_halo_tile(float* __restrict__ tile, const TX* __restrict__ x, int b, int h0, int w0, int c0, int H, int W, int C) {
    // tile[(ih*IW + iw)*CB + c] ; 4-channel vectors, consecutive threads -> consecutive channels
    constexpr int NV = IH * IW * (CB / 4);
    for (int i = threadIdx.x; i < NV; i += 256) {
        const int cv = i % (CB / 4);
        const int pix = i / (CB / 4);
        const int iw = pix % IW, ih = pix / IW;
        const int h = h0 + ih - 3, w = w0 + iw - 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const TX* p = x + (((int64_t)b * H + h) * W + w) * C + c0 + 4 * cv;
            if constexpr (sizeof(TX) == 4) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                const uint2 r = *reinterpret_cast<const uint2*>(p);
                const bf16_t* hh = reinterpret_cast<const bf16_t*>(&r);
                v = make_float4((float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]);
            }
        }
        *reinterpret_cast<float4*>(tile + pix * CB + 4 * cv) = v;
    }
}

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void dwconv7_kernel(const DwP p) {
    __shared__ __attribute__((aligned(16))) float tile[IH * IW * CB];
    int bid = blockIdx.x;
    const int cb = bid % p.cblocks;
    bid /= p.cblocks;
    const int tw = bid % p.tiles_w;
    bid /= p.tiles_w;
    const int th = bid % p.tiles_h;
    const int b = bid / p.tiles_h;
    const int h0 = th * TH, w0 = tw * TW, c0 = cb * CB;

    load_halo_tile<TX>(tile, reinterpret_cast<const TX*>(p.x), b, h0, w0, c0, p.H, p.W, p.C);
    __syncthreads();

    const int cl = threadIdx.x & 31;
    const int r = threadIdx.x >> 5;
    const int c = c0 + cl;
    float acc[TW];
    const float bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[j] = bv;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        float wv[7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const int tap = p.flip ? (6 - ky) * 7 + (6 - kx) : ky * 7 + kx;
            wv[kx] = p.w49[tap * p.C + c];
        }
        float in[IW];
        const float* row = tile + ((r + ky) * IW) * CB + cl;
#pragma unroll
        for (int j = 0; j < IW; ++j) in[j] = row[j * CB];
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) acc[j] = fmaf(wv[kx], in[j + kx], acc[j]);
    }
    const int h = h0 + r;
    if (h >= p.H) return;
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int w = w0 + j;
        if (w < p.W) {
            const int64_t off = (((int64_t)b * p.H + h) * p.W + w) * p.C + c;
            float v = acc[j];
            if (p.res) v += p.res[off];
            reinterpret_cast<TY*>(p.y)[off] = from_f<TY>(v);
        }
    }
}

struct DwWgP {
    const void* x;
    const void* dy;
    float* dw;
    float* db;
    int B, H, W, C;
    int tiles_h, tiles_w, cblocks, ntile;  // ntile = B * tiles_h * tiles_w
};

// dW[c, ky, kx] += sum dY[b,h,w,c] * x[b,h+ky-3,w+kx-3,c];   db[c] += sum dY
// Each workgroup owns one 32-channel block and walks tiles grid-stride, keeping its
// 49 tap sums per thread in registers; one LDS reduction + one atomic per tap at the end.
template <typename TX, typename TDY>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const DwWgP p) {
    __shared__ __attribute__((aligned(16))) float tile[IH * IW * CB];
    __shared__ __attribute__((aligned(16))) float dyt[TH * TW * CB];
    const int cb = blockIdx.x % p.cblocks;
    const int walker = blockIdx.x / p.cblocks;
    const int nwalk = gridDim.x / p.cblocks;
    const int c0 = cb * CB;
    const int cl = threadIdx.x & 31;
    const int r = threadIdx.x >> 5;

    float acc[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = 0.f;
    float accb = 0.f;

    for (int t = walker; t < p.ntile; t += nwalk) {
        const int tw = t % p.tiles_w;
        const int t2 = t / p.tiles_w;
        const int th = t2 % p.tiles_h;
        const int b = t2 / p.tiles_h;
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();  // previous iteration's readers are done
        load_halo_tile<TX>(tile, reinterpret_cast<const TX*>(p.x), b, h0, w0, c0, p.H, p.W, p.C);
        for (int i = threadIdx.x; i < TH * TW * (CB / 4); i += 256) {
            const int cv = i % (CB / 4);
            const int pix = i / (CB / 4);
            const int j = pix % TW, rr = pix / TW;
            const int h = h0 + rr, w = w0 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < p.H && w < p.W) {
                const TDY* q = reinterpret_cast<const TDY*>(p.dy) + (((int64_t)b * p.H + h) * p.W + w) * p.C + c0 + 4 * cv;
                if constexpr (sizeof(TDY) == 4) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    const uint2 rw = *reinterpret_cast<const uint2*>(q);
                    const bf16_t* hh = reinterpret_cast<const bf16_t*>(&rw);
                    v = make_float4((float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]);
                }
            }
            *reinterpret_cast<float4*>(dyt + pix * CB + 4 * cv) = v;
        }
        __syncthreads();
        float d[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            d[j] = dyt[(r * TW + j) * CB + cl];
            accb += d[j];
        }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            float in[IW];
            const float* row = tile + ((r + ky) * IW) * CB + cl;
#pragma unroll
            for (int j = 0; j < IW; ++j) in[j] = row[j * CB];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[ky * 7 + kx] = fmaf(d[j], in[j + kx], acc[ky * 7 + kx]);
        }
    }
    // reduce over the 8 row-threads of each channel through LDS (reuse `tile`)
    __syncthreads();
    float* red = tile;  // [8][50][32] floats = 12800 <= IH*IW*CB = 9856? no -> use two passes
    // pass A: taps 0..24, pass B: taps 25..48 + bias  (8*25*32 = 6400 floats each)
    for (int pass = 0; pass < 2; ++pass) {
        const int tbeg = pass * 25;
        const int tcnt = pass == 0 ? 25 : 24;
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if (t < tcnt) red[(r * 25 + t) * CB + cl] = acc[tbeg + t];
            else if (pass == 1) red[(r * 25 + t) * CB + cl] = accb;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 25 * CB; i += 256) {
            const int t = i / CB, c = i % CB;
            float s = 0.f;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) s += red[(rr * 25 + t) * CB + c];
            if (t < tcnt) atomicAdd(p.dw + (int64_t)(c0 + c) * 49 + tbeg + t, s);
            else if (pass == 1 && p.db) atomicAdd(p.db + c0 + c, s);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int lnx_dwconv7_fwd(const lnx_dwconv_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->w49 && a->y, "lnx_dwconv7_fwd: null operand");
    LNX_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->C % CB == 0, "lnx_dwconv7_fwd: bad shape B=%d H=%d W=%d C=%d (C %% 32)", a->B, a->H, a->W, a->C);
    DwP p;
    p.x = a->x; p.w49 = a->w49; p.bias = a->bias; p.res = a->res; p.y = a->y;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C; p.flip = a->flip;
    p.tiles_h = cdiv(a->H, TH); p.tiles_w = cdiv(a->W, TW); p.cblocks = a->C / CB;
    const int64_t grid = (int64_t)a->B * p.tiles_h * p.tiles_w * p.cblocks;
    LNX_CHECK(grid < (1ll << 31), "lnx_dwconv7_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const int code = a->x_dtype * 2 + a->y_dtype;
    switch (code) {
        case 0: hipLaunchKernelGGL((dwconv7_kernel<float, float>), dim3((int)grid), dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((dwconv7_kernel<float, bf16_t>), dim3((int)grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((dwconv7_kernel<bf16_t, float>), dim3((int)grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((dwconv7_kernel<bf16_t, bf16_t>), dim3((int)grid), dim3(256), 0, st, p); break;
        default: LNX_CHECK(false, "lnx_dwconv7_fwd: bad dtypes");
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_dwconv7_wgrad(const lnx_dwconv_wgrad_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->dy && a->dw, "lnx_dwconv7_wgrad: null operand");
    LNX_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->C % CB == 0, "lnx_dwconv7_wgrad: bad shape");
    DwWgP p;
    p.x = a->x; p.dy = a->dy; p.dw = a->dw; p.db = a->db;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C;
    p.tiles_h = cdiv(a->H, TH); p.tiles_w = cdiv(a->W, TW); p.cblocks = a->C / CB;
    p.ntile = a->B * p.tiles_h * p.tiles_w;
    int walkers = 1024 / p.cblocks;
    if (walkers < 1) walkers = 1;
    if (walkers > p.ntile) walkers = p.ntile;
    const int grid = walkers * p.cblocks;
    hipStream_t st = (hipStream_t)stream;
    const int code = a->x_dtype * 2 + a->dy_dtype;
    switch (code) {
        case 0: hipLaunchKernelGGL((dwconv7_wgrad_kernel<float, float>), dim3(grid), dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL((dwconv7_wgrad_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((dwconv7_wgrad_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((dwconv7_wgrad_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, p); break;
        default: LNX_CHECK(false, "lnx_dwconv7_wgrad: bad dtypes");
    }
    LNX_LAUNCH_CHECK();
    return 0;
}
