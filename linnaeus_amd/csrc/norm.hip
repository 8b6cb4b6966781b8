// LayerNorm over the last (channel) dimension, forward and backward, for gfx950.
//
// One 64-lane wave owns one row: the row lives in registers (<= 2048 channels), statistics
// are two-pass fp32 (mean, then centred variance -- the formulation of nn.LayerNorm and of
// the reference's LayerNormChannelsFirst, blocks/convnext.py:36-38), loads/stores are
// 16-byte vectors.  HBM-bound: algorithmic traffic is one read + one write of the row.
//
// In NHWC (channels-last) memory the reference's channels-first LayerNorm is this same
// kernel, so stem.1 / downsample_layers.*.norm / block norms / norm1 / norm2 / norm_1 /
// norm_2 / final_norm / meta-head norms all map here (eps passed in).
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int MAXV = 8;  // up to 8 float4 per lane: C <= 2048

struct LnP {
    const void* x;
    const void* add;   // optional, type of x, compact rows (row m)
    void* y;
    const float* w;
    const float* b;
    float* mean;
    float* rstd;
    int64_t ldx, ldy, ldadd;
    RowMap xmap, ymap;
    int M, C;
    float eps;
};

template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void store4(T* p, float4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float4 v) {
    uint2 r;
    bf16_t* h = reinterpret_cast<bf16_t*>(&r);
    h[0] = (bf16_t)v.x;
    h[1] = (bf16_t)v.y;
    h[2] = (bf16_t)v.z;
    h[3] = (bf16_t)v.w;
    *reinterpret_cast<uint2*>(p) = r;
}

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnP p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = p.C >> 2;
    const float invC = 1.0f / (float)p.C;
    for (int m = blockIdx.x * 4 + wave; m < p.M; m += gridDim.x * 4) {
        const TX* xr = reinterpret_cast<const TX*>(p.x) + map_row(p.xmap, m) * p.ldx;
        float4 v[MAXV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nvec) {
                v[i] = load4<TX>(xr + 4 * c4);
                sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        const float mu = wave_sum(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nvec) {
                v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
                sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(sq) * invC + p.eps);
        if (lane == 0) {
            if (p.mean) p.mean[m] = mu;
            if (p.rstd) p.rstd[m] = rs;
        }
        TY* yr = reinterpret_cast<TY*>(p.y) + map_row(p.ymap, m) * p.ldy;
        const TX* ar = p.add ? reinterpret_cast<const TX*>(p.add) + (int64_t)m * p.ldadd : nullptr;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nvec) {
                const float4 w = *reinterpret_cast<const float4*>(p.w + 4 * c4);
                const float4 b = *reinterpret_cast<const float4*>(p.b + 4 * c4);
                float4 o = make_float4(v[i].x * rs * w.x + b.x, v[i].y * rs * w.y + b.y, v[i].z * rs * w.z + b.z, v[i].w * rs * w.w + b.w);
                if (ar) {
                    const float4 a = load4<TX>(ar + 4 * c4);
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                store4<TY>(yr + 4 * c4, o);
            }
        }
    }
}

struct LnBwdP {
    const void* dy;   // [M, C] rows via dymap
    const void* x;    // forward input, rows via xmap
    const float* w;
    const float* mean;
    const float* rstd;
    const float* gin; // optional fp32 gradient already flowing into x (rows via xmap, ld = ldgin)
    void* dx;         // output rows via xmap (ld = lddx)
    float* dw;        // [C] atomics
    float* db;        // [C] atomics
    int64_t lddy, ldx, lddx, ldgin;
    RowMap dymap, xmap;
    int M, C;
    int relu_mask;
};

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (x - mean) * rstd
// dw += sum_m dy * xhat,  db += sum_m dy
template <typename TDY, typename TX, typename TDX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdP p) {
    __shared__ float red[4][64 * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = p.C >> 2;
    const float invC = 1.0f / (float)p.C;
    float4 adw[MAXV], adb[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) adw[i] = adb[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int m = blockIdx.x * 4 + wave; m < p.M; m += gridDim.x * 4) {
        const int64_t xrow = map_row(p.xmap, m);
        const TX* xr = reinterpret_cast<const TX*>(p.x) + xrow * p.ldx;
        const TDY* dr = reinterpret_cast<const TDY*>(p.dy) + map_row(p.dymap, m) * p.lddy;
        const float mu = p.mean[m], rs = p.rstd[m];
        float4 xh[MAXV], g[MAXV];
        unsigned pos[MAXV];  // bit j: x element j > 0 (ReLU mask)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nvec) {
                const float4 xv = load4<TX>(xr + 4 * c4);
                const float4 dv = load4<TDY>(dr + 4 * c4);
                pos[i] = (xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u);
                const float4 w = *reinterpret_cast<const float4*>(p.w + 4 * c4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * w.x, dv.y * w.y, dv.z * w.z, dv.w * w.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                adw[i].x += dv.x * xh[i].x; adw[i].y += dv.y * xh[i].y; adw[i].z += dv.z * xh[i].z; adw[i].w += dv.w * xh[i].w;
                adb[i].x += dv.x; adb[i].y += dv.y; adb[i].z += dv.z; adb[i].w += dv.w;
            }
        }
        const float m1 = wave_sum(s1) * invC;
        const float m2 = wave_sum(s2) * invC;
        TDX* ox = reinterpret_cast<TDX*>(p.dx) + xrow * p.lddx;
        const float* gi = p.gin ? p.gin + xrow * p.ldgin : nullptr;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nvec) {
                float4 o = make_float4(rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2),
                                       rs * (g[i].z - m1 - xh[i].z * m2), rs * (g[i].w - m1 - xh[i].w * m2));
                if (gi) {
                    const float4 a = *reinterpret_cast<const float4*>(gi + 4 * c4);
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                if (p.relu_mask) {
                    if (!(pos[i] & 1u)) o.x = 0.f;
                    if (!(pos[i] & 2u)) o.y = 0.f;
                    if (!(pos[i] & 4u)) o.z = 0.f;
                    if (!(pos[i] & 8u)) o.w = 0.f;
                }
                store4<TDX>(ox + 4 * c4, o);
            }
        }
    }
    if (p.dw == nullptr && p.db == nullptr) return;
    // reduce the four waves' partial column sums through LDS, one atomic per column per block
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? p.dw : p.db;
        if (dst == nullptr) continue;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (64 * i >= nvec) break;
            const float4 a = pass == 0 ? adw[i] : adb[i];
            __syncthreads();
            *reinterpret_cast<float4*>(&red[wave][lane * 4]) = a;
            __syncthreads();
            if (wave == 0) {
                const int c4 = lane + 64 * i;
                if (c4 < nvec) {
                    float4 t = *reinterpret_cast<float4*>(&red[0][lane * 4]);
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) {
                        const float4 u = *reinterpret_cast<float4*>(&red[ww][lane * 4]);
                        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                    }
                    atomicAdd(dst + 4 * c4 + 0, t.x);
                    atomicAdd(dst + 4 * c4 + 1, t.y);
                    atomicAdd(dst + 4 * c4 + 2, t.z);
                    atomicAdd(dst + 4 * c4 + 3, t.w);
                }
            }
        }
    }
}

inline int ln_grid(int M) {
    const int wg = cdiv(M, 4);
    return wg < 2048 ? wg : 2048;
}

}  // namespace

extern "C" int lnx_layernorm_fwd(const lnx_ln_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->y && a->w && a->b, "lnx_layernorm_fwd: null operand");
    LNX_CHECK(a->M > 0 && a->C > 0 && a->C % 4 == 0 && a->C <= 64 * 4 * MAXV, "lnx_layernorm_fwd: bad shape M=%d C=%d", a->M, a->C);
    LNX_CHECK(a->ldx % 4 == 0 && a->ldy % 4 == 0, "lnx_layernorm_fwd: ldx/ldy must be multiples of 4");
    LnP p;
    p.x = a->x; p.add = a->add; p.y = a->y; p.w = a->w; p.b = a->b; p.mean = a->mean; p.rstd = a->rstd;
    p.ldx = a->ldx; p.ldy = a->ldy; p.ldadd = a->ldadd;
    p.xmap = RowMap{a->x_map.group, a->x_map.pad, a->x_map.off};
    p.ymap = RowMap{a->y_map.group, a->y_map.pad, a->y_map.off};
    p.M = a->M; p.C = a->C; p.eps = a->eps;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ln_grid(a->M)), block(256);
    const int xi = a->x_dtype, yi = a->y_dtype;
    if (xi == LNX_F32 && yi == LNX_F32) hipLaunchKernelGGL((ln_fwd_kernel<float, float>), grid, block, 0, st, p);
    else if (xi == LNX_F32 && yi == LNX_BF16) hipLaunchKernelGGL((ln_fwd_kernel<float, bf16_t>), grid, block, 0, st, p);
    else if (xi == LNX_BF16 && yi == LNX_F32) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, float>), grid, block, 0, st, p);
    else if (xi == LNX_BF16 && yi == LNX_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, st, p);
    else LNX_CHECK(false, "lnx_layernorm_fwd: bad dtypes %d %d", xi, yi);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_layernorm_bwd(const lnx_ln_bwd_args* a, void* stream) {
    LNX_CHECK(a && a->dy && a->x && a->w && a->mean && a->rstd && a->dx, "lnx_layernorm_bwd: null operand");
    LNX_CHECK(a->M > 0 && a->C > 0 && a->C % 4 == 0 && a->C <= 64 * 4 * MAXV, "lnx_layernorm_bwd: bad shape M=%d C=%d", a->M, a->C);
    LNX_CHECK(a->ldx % 4 == 0 && a->lddy % 4 == 0 && a->lddx % 4 == 0, "lnx_layernorm_bwd: leading dims must be multiples of 4");
    LnBwdP p;
    p.dy = a->dy; p.x = a->x; p.w = a->w; p.mean = a->mean; p.rstd = a->rstd; p.gin = a->gin; p.dx = a->dx; p.dw = a->dw; p.db = a->db;
    p.lddy = a->lddy; p.ldx = a->ldx; p.lddx = a->lddx; p.ldgin = a->ldgin;
    p.dymap = RowMap{a->dy_map.group, a->dy_map.pad, a->dy_map.off};
    p.xmap = RowMap{a->x_map.group, a->x_map.pad, a->x_map.off};
    p.M = a->M; p.C = a->C; p.relu_mask = a->relu_mask;
    hipStream_t st = (hipStream_t)stream;
    int g = cdiv(a->M, 4);
    if (g > 1024) g = 1024;
    const dim3 grid(g), block(256);
    const int code = a->dy_dtype * 4 + a->x_dtype * 2 + a->dx_dtype;
#define LNB(TDY, TX, TDX) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX>), grid, block, 0, st, p)
    switch (code) {
        case 0: LNB(float, float, float); break;
        case 1: LNB(float, float, bf16_t); break;
        case 2: LNB(float, bf16_t, float); break;
        case 3: LNB(float, bf16_t, bf16_t); break;
        case 4: LNB(bf16_t, float, float); break;
        case 5: LNB(bf16_t, float, bf16_t); break;
        case 6: LNB(bf16_t, bf16_t, float); break;
        case 7: LNB(bf16_t, bf16_t, bf16_t); break;
        default: LNX_CHECK(false, "lnx_layernorm_bwd: bad dtypes");
    }
#undef LNB
    LNX_LAUNCH_CHECK();
    return 0;
}
