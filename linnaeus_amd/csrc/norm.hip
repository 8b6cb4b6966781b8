// LayerNorm over the last (channel) dimension, forward and backward, for gfx950.
//
// A row is owned by a group of G lanes (G = 8/16/32/64 chosen from C so that each lane holds
// <= 3 float4; C up to 2048 uses a full wave with up to 8), i.e. a 64-lane wave processes
// 64/G consecutive rows at once: for the ConvNeXt widths (C = 96/192) every lane is busy and a
// wave-instruction touches 1 KiB of consecutive memory.  The row lives in registers,
// statistics are two-pass fp32 (mean, then centred variance -- the formulation of
// nn.LayerNorm and of the reference's LayerNormChannelsFirst, blocks/convnext.py:36-38),
// loads/stores are 16-byte (fp32) / 8-byte (bf16) vectors.  HBM-bound: algorithmic traffic is
// one read + one write of the row.
//
// In NHWC (channels-last) memory the reference's channels-first LayerNorm is this same
// kernel, so stem.1 / downsample_layers.*.norm / block norms / norm1 / norm2 / norm_1 /
// norm_2 / final_norm / meta-head norms all map here (eps passed in).
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int MAXC = 2048;

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
    unsigned char* y8 = nullptr;   // optional MXFP8 copy of the bf16 output (ln_fwd_kernel<..., MX = true>)
    unsigned char* y8s = nullptr;  // its block scales, [C/128][M][4]
    int64_t ldy8 = 0;
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

// Column index (in float4 units) of slot i of lane `sub`.  PAIR = 1: lanes interleave single float4 (8-byte accesses for
// bf16 rows).  PAIR = 2: a lane owns adjacent float4 pairs, so a bf16 row is read / written in 16-byte pieces (8-byte
// accesses run at 0.54-0.70x the 16-byte rate); used when the streamed operands are bf16.
template <int G, int PAIR> __device__ __forceinline__ int cidx(int sub, int i) { return PAIR == 1 ? sub + G * i : 2 * sub + (i & 1) + 2 * G * (i >> 1); }

template <typename T> __device__ __forceinline__ void load4x2(const T* p, float4& a, float4& b);
template <> __device__ __forceinline__ void load4x2<float>(const float* p, float4& a, float4& b) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
}
template <> __device__ __forceinline__ void load4x2<bf16_t>(const bf16_t* p, float4& a, float4& b) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
    a = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    b = make_float4((float)h[4], (float)h[5], (float)h[6], (float)h[7]);
}
template <typename T> __device__ __forceinline__ void store4x2(T* p, float4 a, float4 b);
template <> __device__ __forceinline__ void store4x2<float>(float* p, float4 a, float4 b) {
    *reinterpret_cast<float4*>(p) = a;
    *reinterpret_cast<float4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store4x2<bf16_t>(bf16_t* p, float4 a, float4 b) {
    uint4 r;
    bf16_t* h = reinterpret_cast<bf16_t*>(&r);
    h[0] = (bf16_t)a.x; h[1] = (bf16_t)a.y; h[2] = (bf16_t)a.z; h[3] = (bf16_t)a.w;
    h[4] = (bf16_t)b.x; h[5] = (bf16_t)b.y; h[6] = (bf16_t)b.z; h[7] = (bf16_t)b.w;
    *reinterpret_cast<uint4*>(p) = r;
}

template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// FULL: C / 4 == G * V, every slot of every lane is a column (all the model's widths).  The loads are unconditional
// either way (a slot beyond the row reads column 0 and is zeroed afterwards): a load under a branch whose result is
// merged with a zero is waited for on the spot (s_waitcnt vmcnt(0) right behind it), which turns the V loads of a row
// into V serial round trips.  gamma / beta live in registers for the whole kernel for the same reason.
// MX (PAIR == 1, bf16 output): the row is also written as MXFP8 -- the bf16-rounded output quantised exactly as
// lnx_quantize_mxfp8 does it.  A 32-element block is the float4 slot i of 8 consecutive lanes: the block maximum is three
// DPP steps (two quad permutes and a half-row mirror), each lane converts its own 4 values into one dword.
__device__ __forceinline__ float max8(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));  // row_half_mirror
    return v;
}

template <typename TX, typename TY, int G, int V, int PAIR = 1, bool FULL = false, bool MX = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnP p) {
    static_assert(!MX || (PAIR == 1 && G % 8 == 0 && sizeof(TY) == 2), "MX output: 8 consecutive lanes hold one 32-element block");
    constexpr int R = 64 / G;  // rows per wave
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane % G, rw = lane / G;
    const int nvec = p.C >> 2;
    const float invC = 1.0f / (float)p.C;
    int col[V];   // first element of slot i (clamped to 0 beyond the row)
    bool ok[V];
    float4 wv[V], bv[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c4 = cidx<G, PAIR>(sub, i);
        ok[i] = FULL || c4 < nvec;
        col[i] = ok[i] ? 4 * c4 : 0;
        wv[i] = *reinterpret_cast<const float4*>(p.w + col[i]);
        bv[i] = *reinterpret_cast<const float4*>(p.b + col[i]);
    }
    for (int m0 = (blockIdx.x * 4 + wave) * R; m0 < p.M; m0 += gridDim.x * 4 * R) {
        const int m = m0 + rw;
        const bool rv = m < p.M;
        const int mc = rv ? m : p.M - 1;
        const TX* xr = reinterpret_cast<const TX*>(p.x) + map_row(p.xmap, mc) * p.ldx;
        float4 v[V], av[V];
#pragma unroll
        for (int i = 0; i < V; i += PAIR) {
            if constexpr (PAIR == 2) load4x2<TX>(xr + col[i], v[i], v[i + 1]);
            else v[i] = load4<TX>(xr + col[i]);
        }
        if (p.add) {  // uniform
            const TX* ar = reinterpret_cast<const TX*>(p.add) + (int64_t)mc * p.ldadd;
#pragma unroll
            for (int i = 0; i < V; i += PAIR) {
                if constexpr (PAIR == 2) load4x2<TX>(ar + col[i], av[i], av[i + 1]);
                else av[i] = load4<TX>(ar + col[i]);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (!ok[i]) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = group_sum<G>(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            if (ok[i]) {
                v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
                sq += fmaf(v[i].x, v[i].x, v[i].y * v[i].y) + fmaf(v[i].z, v[i].z, v[i].w * v[i].w);  // contraction spelled out, as below
            }
        }
        const float rs = 1.0f / sqrtf(fmaf(group_sum<G>(sq), invC, p.eps));
        if (!rv) continue;
        if (sub == 0) {
            if (p.mean) p.mean[m] = mu;
            if (p.rstd) p.rstd[m] = rs;
        }
        TY* yr = reinterpret_cast<TY*>(p.y) + map_row(p.ymap, m) * p.ldy;
#pragma unroll
        for (int i = 0; i < V; i += PAIR) {
            float4 o[PAIR];
            if (ok[i]) {
#pragma unroll
                for (int q = 0; q < PAIR; ++q) {
                    const float4 w = wv[i + q], b = bv[i + q];
                    // explicit fma: every instantiation (with / without the MXFP8 copy, full / partial rows) rounds the same way
                    o[q] = make_float4(fmaf(v[i + q].x * rs, w.x, b.x), fmaf(v[i + q].y * rs, w.y, b.y), fmaf(v[i + q].z * rs, w.z, b.z), fmaf(v[i + q].w * rs, w.w, b.w));
                    if (p.add) {
                        o[q].x += av[i + q].x; o[q].y += av[i + q].y; o[q].z += av[i + q].z; o[q].w += av[i + q].w;
                    }
                }
                if constexpr (PAIR == 2) store4x2<TY>(yr + col[i], o[0], o[1]);
                else store4<TY>(yr + col[i], o[0]);
            }
            if constexpr (MX) {
                // every lane takes part in the block maximum (a slot beyond the row contributes zeros and stores nothing)
                float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
                if (ok[i]) {
                    q0 = (float)(bf16_t)o[0].x; q1 = (float)(bf16_t)o[0].y; q2 = (float)(bf16_t)o[0].z; q3 = (float)(bf16_t)o[0].w;
                }
                const float am = max8(fmaxf(fmaxf(fabsf(q0), fabsf(q1)), fmaxf(fabsf(q2), fabsf(q3))));
                const uint32_t bits = __float_as_uint(am);
                int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                e = e < 0 ? 0 : (e > 254 ? 254 : e);
                const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(q0 * inv, -448.f), 448.f), fminf(fmaxf(q1 * inv, -448.f), 448.f), 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(q2 * inv, -448.f), 448.f), fminf(fmaxf(q3 * inv, -448.f), 448.f), pk, true);
                if (ok[i]) {
                    *reinterpret_cast<int*>(p.y8 + (int64_t)m * p.ldy8 + col[i]) = pk;
                    const int kb = col[i] >> 5;
                    if ((sub & 7) == 0) p.y8s[((int64_t)(kb >> 2) * p.M + m) * 4 + (kb & 3)] = (unsigned char)e;
                }
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
    float* part;      // optional workspace [gridDim.x][2][C]: per-workgroup column partials (no atomics)
    void* dx2;        // optional second output: rs2[m / rps2] * dx, identity rows, ld = lddx2, bf16 or fp32
    const float* rs2;
    int64_t lddx2;
    int rps2, dx2_f32;
    unsigned char* dx2_8;   // optional MXFP8 copy of the bf16 dx2 (lnx_ln_bwd_args.dx2_8): elements, E8M0 block scales
    unsigned char* dx2_8s;
    int64_t lddx2_8;
};

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w,  xhat = (x - mean) * rstd
// dw += sum_m dy * xhat,  db += sum_m dy
template <typename TDY, typename TX, typename TDX, int G, int V, int PAIR = 1, bool FULL = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdP p) {
    constexpr int R = 64 / G;
    __shared__ float red[4][G * V * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane % G, rw = lane / G;
    const int nvec = p.C >> 2;
    const float invC = 1.0f / (float)p.C;
    float4 adw[V], adb[V], wv[V];
    int col[V];   // first element of slot i (clamped to 0 beyond the row); loads are unconditional, see ln_fwd_kernel
    bool ok[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        adw[i] = adb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c4 = cidx<G, PAIR>(sub, i);
        ok[i] = FULL || c4 < nvec;
        col[i] = ok[i] ? 4 * c4 : 0;
        wv[i] = *reinterpret_cast<const float4*>(p.w + col[i]);
    }

    for (int m0 = (blockIdx.x * 4 + wave) * R; m0 < p.M; m0 += gridDim.x * 4 * R) {
        const int m = m0 + rw;
        const bool rv = m < p.M;
        const int mc = rv ? m : p.M - 1;
        const int64_t xrow = map_row(p.xmap, mc);
        const TX* xr = reinterpret_cast<const TX*>(p.x) + xrow * p.ldx;
        const TDY* dr = reinterpret_cast<const TDY*>(p.dy) + map_row(p.dymap, mc) * p.lddy;
        const float mu = p.mean[mc], rs = p.rstd[mc];
        float4 xh[V], g[V];
        unsigned pos[V];  // bit j: x element j > 0 (ReLU mask)
        float s1 = 0.f, s2 = 0.f;
        float4 xraw[V], draw[V], gv[V];
#pragma unroll
        for (int i = 0; i < V; i += PAIR) {
            if constexpr (PAIR == 2) {
                load4x2<TX>(xr + col[i], xraw[i], xraw[i + 1]);
                load4x2<TDY>(dr + col[i], draw[i], draw[i + 1]);
            } else {
                xraw[i] = load4<TX>(xr + col[i]);
                draw[i] = load4<TDY>(dr + col[i]);
            }
        }
        if (p.gin) {  // uniform
            const float* gi = p.gin + xrow * p.ldgin;
#pragma unroll
            for (int i = 0; i < V; ++i) gv[i] = *reinterpret_cast<const float4*>(gi + col[i]);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            pos[i] = 0;
            if (ok[i]) {
                const float4 xv = xraw[i];
                float4 dv = draw[i];
                if (!rv) dv = make_float4(0.f, 0.f, 0.f, 0.f);
                pos[i] = (xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * wv[i].x, dv.y * wv[i].y, dv.z * wv[i].z, dv.w * wv[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                adw[i].x += dv.x * xh[i].x; adw[i].y += dv.y * xh[i].y; adw[i].z += dv.z * xh[i].z; adw[i].w += dv.w * xh[i].w;
                adb[i].x += dv.x; adb[i].y += dv.y; adb[i].z += dv.z; adb[i].w += dv.w;
            }
        }
        const float m1 = group_sum<G>(s1) * invC;
        const float m2 = group_sum<G>(s2) * invC;
        if (!rv) continue;
        TDX* ox = reinterpret_cast<TDX*>(p.dx) + xrow * p.lddx;
#pragma unroll
        for (int i = 0; i < V; i += PAIR) {
            const int c4 = col[i] >> 2;
            float4 q8 = make_float4(0.f, 0.f, 0.f, 0.f);  // the bf16-rounded dx2 values of this slot (MXFP8 copy; zeros beyond the row)
            if (ok[i]) {
                float4 ov[PAIR];
#pragma unroll
                for (int q = 0; q < PAIR; ++q) {
                    const int k = i + q;
                    float4 o = make_float4(rs * (g[k].x - m1 - xh[k].x * m2), rs * (g[k].y - m1 - xh[k].y * m2),
                                           rs * (g[k].z - m1 - xh[k].z * m2), rs * (g[k].w - m1 - xh[k].w * m2));
                    if (p.gin) {
                        const float4 a = gv[k];
                        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                    }
                    if (p.relu_mask) {
                        if (!(pos[k] & 1u)) o.x = 0.f;
                        if (!(pos[k] & 2u)) o.y = 0.f;
                        if (!(pos[k] & 4u)) o.z = 0.f;
                        if (!(pos[k] & 8u)) o.w = 0.f;
                    }
                    ov[q] = o;
                }
                if constexpr (PAIR == 2) store4x2<TDX>(ox + 4 * c4, ov[0], ov[1]);
                else store4<TDX>(ox + 4 * c4, ov[0]);
                if (p.dx2) {  // DropPath-scaled copy in storage type for the next branch's GEMMs (wave-uniform branch)
                    const float sc = p.rs2 ? p.rs2[m / p.rps2] : 1.0f;
#pragma unroll
                    for (int q = 0; q < PAIR; ++q) ov[q] = make_float4(ov[q].x * sc, ov[q].y * sc, ov[q].z * sc, ov[q].w * sc);
                    if (p.dx2_f32) {
                        float* o2 = reinterpret_cast<float*>(p.dx2) + (int64_t)m * p.lddx2 + 4 * c4;
#pragma unroll
                        for (int q = 0; q < PAIR; ++q) store4<float>(o2 + 4 * q, ov[q]);
                    } else {
                        bf16_t* o2 = reinterpret_cast<bf16_t*>(p.dx2) + (int64_t)m * p.lddx2 + 4 * c4;
                        if constexpr (PAIR == 2) store4x2<bf16_t>(o2, ov[0], ov[1]);
                        else store4<bf16_t>(o2, ov[0]);
                        q8 = make_float4((float)(bf16_t)ov[0].x, (float)(bf16_t)ov[0].y, (float)(bf16_t)ov[0].z, (float)(bf16_t)ov[0].w);
                    }
                }
            }
            if constexpr (PAIR == 1 && G % 8 == 0) {
                if (p.dx2_8) {  // kernel-uniform.  ln_fwd_kernel's MX block on the stored (rounded) values: 8 consecutive lanes = one 32-element block
                    const float am = max8(fmaxf(fmaxf(fabsf(q8.x), fabsf(q8.y)), fmaxf(fabsf(q8.z), fabsf(q8.w))));
                    const uint32_t bits = __float_as_uint(am);
                    int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                    e = e < 0 ? 0 : (e > 254 ? 254 : e);
                    const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(q8.x * inv, -448.f), 448.f), fminf(fmaxf(q8.y * inv, -448.f), 448.f), 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(q8.z * inv, -448.f), 448.f), fminf(fmaxf(q8.w * inv, -448.f), 448.f), pk, true);
                    if (ok[i]) {
                        *reinterpret_cast<int*>(p.dx2_8 + (int64_t)m * p.lddx2_8 + col[i]) = pk;
                        const int kb = col[i] >> 5;
                        if ((sub & 7) == 0) p.dx2_8s[((int64_t)(kb >> 2) * p.M + m) * 4 + (kb & 3)] = (unsigned char)e;
                    }
                }
            }
        }
    }
    if (p.dw == nullptr && p.db == nullptr) return;
    // column partials: sum the 64/G row groups of the wave (shuffles), then the 4 waves (LDS),
    // then one atomic per column per workgroup
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? p.dw : p.db;
        if (dst == nullptr) continue;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float4 a = pass == 0 ? adw[i] : adb[i];
#pragma unroll
            for (int o = G; o < 64; o <<= 1) {
                a.x += __shfl_xor(a.x, o, 64);
                a.y += __shfl_xor(a.y, o, 64);
                a.z += __shfl_xor(a.z, o, 64);
                a.w += __shfl_xor(a.w, o, 64);
            }
            if (rw == 0) *reinterpret_cast<float4*>(&red[wave][(i * G + sub) * 4]) = a;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < G * V; e += 256) {
            const int i = e / G, sb = e % G;
            const int c4 = cidx<G, PAIR>(sb, i);
            if (c4 < nvec) {
                float4 t = *reinterpret_cast<float4*>(&red[0][e * 4]);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) {
                    const float4 u = *reinterpret_cast<float4*>(&red[ww][e * 4]);
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                if (p.part) {
                    *reinterpret_cast<float4*>(p.part + ((int64_t)blockIdx.x * 2 + pass) * p.C + 4 * c4) = t;
                } else {
                    atomicAdd(dst + 4 * c4 + 0, t.x);
                    atomicAdd(dst + 4 * c4 + 1, t.y);
                    atomicAdd(dst + 4 * c4 + 2, t.z);
                    atomicAdd(dst + 4 * c4 + 3, t.w);
                }
            }
        }
    }
}

// second stage of the dw/db reduction: sums the per-workgroup partials of a slice of workgroups
// (blockIdx.y) for 256 consecutive (pass, column) entries; one atomic per entry per slice
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nwg, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // entry in [0, 2C): pass * C + col
    if (e >= 2 * C) return;
    const int per = (nwg + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;  // independent chains keep 4+ loads in flight
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        t0 += part[(int64_t)w * 2 * C + e];
        t1 += part[(int64_t)(w + 1) * 2 * C + e];
        t2 += part[(int64_t)(w + 2) * 2 * C + e];
        t3 += part[(int64_t)(w + 3) * 2 * C + e];
    }
    for (; w < w1; ++w) t0 += part[(int64_t)w * 2 * C + e];
    const float t = (t0 + t1) + (t2 + t3);
    float* dst = e < C ? dw : db;
    if (dst) atomicAdd(dst + (e < C ? e : e - C), t);
}

// Round 5: the second stages of several LayerNorm backward calls in ONE launch (lnx_ln_bwd_args.defer / lnx_layernorm_bwd_flush): 22 launches
// of ~8 us per training step (1.6 % of a 128-image step) become one per backward segment.  Same arithmetic per entry as ln_bwd_reduce_kernel.
constexpr int LN_BATCH = 16;
struct LnReduceDesc {
    const float* part;
    float* dw;
    float* db;
    int nwg, C, slices, block_start;
};
struct LnReduceBatch {
    LnReduceDesc d[LN_BATCH];
    int n;
};
__global__ __launch_bounds__(256) void ln_bwd_reduce_batch_kernel(const LnReduceBatch b) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < LN_BATCH; ++i)
        if (i < b.n && (int)blockIdx.x >= b.d[i].block_start) j = i;
    const LnReduceDesc& d = b.d[j];
    const int local = (int)blockIdx.x - d.block_start;
    const int nbx = (2 * d.C + 255) / 256;
    const int bx = local % nbx, sl = local / nbx;
    const int e = bx * 256 + threadIdx.x;  // entry in [0, 2C): pass * C + col
    if (e >= 2 * d.C) return;
    const int per = (d.nwg + d.slices - 1) / d.slices;
    const int w0 = sl * per, w1 = min(d.nwg, w0 + per);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        t0 += d.part[(int64_t)w * 2 * d.C + e];
        t1 += d.part[(int64_t)(w + 1) * 2 * d.C + e];
        t2 += d.part[(int64_t)(w + 2) * 2 * d.C + e];
        t3 += d.part[(int64_t)(w + 3) * 2 * d.C + e];
    }
    for (; w < w1; ++w) t0 += d.part[(int64_t)w * 2 * d.C + e];
    const float t = (t0 + t1) + (t2 + t3);
    float* dst = e < d.C ? d.dw : d.db;
    if (dst) atomicAdd(dst + (e < d.C ? e : e - d.C), t);
}

// postponed second stages of this host thread (one training loop = one thread; a call postponed on another stream flushes what is pending first)
static thread_local LnReduceBatch g_ln_pending = {};
static thread_local int g_ln_pending_blocks = 0;
static thread_local hipStream_t g_ln_pending_stream = nullptr;

int ln_flush(hipStream_t st) {
    if (g_ln_pending.n == 0) return 0;
    if (st != nullptr && st != g_ln_pending_stream) return 1;
    hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((unsigned)g_ln_pending_blocks), dim3(256), 0, g_ln_pending_stream, g_ln_pending);
    g_ln_pending.n = 0;
    g_ln_pending_blocks = 0;
    return 0;
}
int ln_discard() {
    const int n = g_ln_pending.n;
    g_ln_pending.n = 0;
    g_ln_pending_blocks = 0;
    g_ln_pending_stream = nullptr;
    return n;
}
}  // namespace
void ln_postpone_reduce(const float* part, int nwg, int C, float* dw, float* db, int slices, hipStream_t st) {
    if (g_ln_pending.n > 0 && (g_ln_pending_stream != st || g_ln_pending.n == LN_BATCH)) (void)ln_flush(nullptr);
    LnReduceDesc& d = g_ln_pending.d[g_ln_pending.n++];
    d.part = part; d.dw = dw; d.db = db; d.nwg = nwg; d.C = C; d.slices = slices; d.block_start = g_ln_pending_blocks;
    g_ln_pending_blocks += ((2 * C + 255) / 256) * slices;
    g_ln_pending_stream = st;
}
namespace {

// (G, V) from C: smallest lane group whose lanes hold <= 3 float4; full wave with 8 otherwise
inline void pick_gv(int C, int& G, int& V) {
    const int nvec = C / 4;
    for (int g = 8; g <= 64; g <<= 1)
        if ((nvec + g - 1) / g <= 3) {
            G = g;
            V = 3;
            return;
        }
    // wide rows: 64 lanes with 4, 6 or 8 float4 slots each (C = 1024 and 1536 fill 4 / 6 exactly: the xl / lg RoPE widths)
    G = 64;
    V = nvec <= 64 * 4 ? 4 : (nvec <= 64 * 6 ? 6 : 8);
}

// pair mode (16-byte accesses on bf16 rows): lane group of G2 lanes, each holding <= 3 float4 pairs; 0 = not applicable
inline int pick_pair_g(int C) {
    if (C % 8 != 0) return 0;
    const int npair = C / 8;
    for (int g = 4; g <= 32; g <<= 1)
        if ((npair + g - 1) / g <= 3) return g;
    return 0;
}

template <typename TX, typename TY>
void launch_fwd(const LnP& p, hipStream_t st) {
    int G, V;
    pick_gv(p.C, G, V);
    if constexpr (sizeof(TX) == 2) {
        const int g2 = pick_pair_g(p.C);
        if (g2 && (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.add) & 15) == 0 && (p.ldx * 2) % 16 == 0 && (p.ldy * sizeof(TY)) % 16 == 0 && (p.ldadd * 2) % 16 == 0) {
            int grid = cdiv(p.M, 4 * (64 / g2));
            if (grid > 4096) grid = 4096;
            const dim3 gg(grid), bb(256);
            if (g2 == 4) { if (p.C / 4 == 4 * 6) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 4, 6, 2, true>), gg, bb, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 4, 6, 2, false>), gg, bb, 0, st, p); }
            else if (g2 == 8) { if (p.C / 4 == 8 * 6) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 8, 6, 2, true>), gg, bb, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 8, 6, 2, false>), gg, bb, 0, st, p); }
            else if (g2 == 16) { if (p.C / 4 == 16 * 6) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 16, 6, 2, true>), gg, bb, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 16, 6, 2, false>), gg, bb, 0, st, p); }
            else { if (p.C / 4 == 32 * 6) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 32, 6, 2, true>), gg, bb, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 32, 6, 2, false>), gg, bb, 0, st, p); }
            return;
        }
    }
    const int rows_per_wg = 4 * (64 / G);
    int grid = cdiv(p.M, rows_per_wg);
    if (grid > 4096) grid = 4096;
    const dim3 g(grid), b(256);
    if constexpr (sizeof(TX) == 4 && sizeof(TY) == 2) {
        if (p.y8) {
            const bool full = p.C / 4 == G * V;
#define LN_MX(GG, VV)                                                                                            \
    do {                                                                                                         \
        if (full) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, GG, VV, 1, true, true>), g, b, 0, st, p);            \
        else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, GG, VV, 1, false, true>), g, b, 0, st, p);                \
    } while (0)
            if (V == 8) LN_MX(64, 8);
            else if (V == 4) LN_MX(64, 4);
            else if (V == 6) LN_MX(64, 6);
            else if (G == 8) LN_MX(8, 3);
            else if (G == 16) LN_MX(16, 3);
            else if (G == 32) LN_MX(32, 3);
            else LN_MX(64, 3);
#undef LN_MX
            return;
        }
    }
    if (V == 8) { if (p.C / 4 == 64 * 8) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 8, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 8, 1, false>), g, b, 0, st, p); }
    else if (V == 4) { if (p.C / 4 == 64 * 4) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 4, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 4, 1, false>), g, b, 0, st, p); }
    else if (V == 6) { if (p.C / 4 == 64 * 6) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 6, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 6, 1, false>), g, b, 0, st, p); }
    else if (G == 8) { if (p.C / 4 == 8 * 3) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 8, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 8, 3, 1, false>), g, b, 0, st, p); }
    else if (G == 16) { if (p.C / 4 == 16 * 3) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 16, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 16, 3, 1, false>), g, b, 0, st, p); }
    else if (G == 32) { if (p.C / 4 == 32 * 3) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 32, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 32, 3, 1, false>), g, b, 0, st, p); }
    else { if (p.C / 4 == 64 * 3) hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_fwd_kernel<TX, TY, 64, 3, 1, false>), g, b, 0, st, p); }
}

constexpr int LN_WS_WGS = 2048;  // workgroups the partial-sum workspace is sized for

template <typename TDY, typename TX, typename TDX>
void launch_bwd(LnBwdP p, hipStream_t st, int64_t ws_floats, bool defer) {
    int G, V;
    pick_gv(p.C, G, V);
    const int rows_per_wg = 4 * (64 / G);
    // same-address float atomics serialise (~0.1 us each): with a workspace every workgroup stores its
    // 2C column partials and a second tiny kernel sums them; without one the grid is kept at ~2
    // workgroups per CU so that few workgroups contend
    int grid = cdiv(p.M, rows_per_wg * 2);
    const bool need_cols = p.dw != nullptr || p.db != nullptr;
    int cap = 512;
    if (p.part && need_cols) {
        cap = (int)(ws_floats / (2 * (int64_t)p.C));
        if (cap > LN_WS_WGS) cap = LN_WS_WGS;
        if (cap < 1) {
            p.part = nullptr;
            cap = 512;
        }
    } else {
        p.part = nullptr;
        if (!need_cols) cap = 4096;
    }
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    const dim3 g(grid), b(256);
    bool launched = false;
    if constexpr (sizeof(TDY) == 2 && sizeof(TX) == 2) {
        const int g2 = pick_pair_g(p.C);
        if (g2 && (((uintptr_t)p.x | (uintptr_t)p.dy | (uintptr_t)p.dx) & 15) == 0 && (p.ldx * 2) % 16 == 0 && (p.lddy * 2) % 16 == 0 && (p.lddx * sizeof(TDX)) % 16 == 0) {
            // same grid: the partial-sum workspace is indexed by workgroup, rows are walked grid-stride
            if (g2 == 4) { if (p.C / 4 == 4 * 6) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 4, 6, 2, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 4, 6, 2, false>), g, b, 0, st, p); }
            else if (g2 == 8) { if (p.C / 4 == 8 * 6) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 8, 6, 2, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 8, 6, 2, false>), g, b, 0, st, p); }
            else if (g2 == 16) { if (p.C / 4 == 16 * 6) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 16, 6, 2, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 16, 6, 2, false>), g, b, 0, st, p); }
            else { if (p.C / 4 == 32 * 6) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 32, 6, 2, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 32, 6, 2, false>), g, b, 0, st, p); }
            launched = true;
        }
    }
    if (launched) {
    } else if (V == 8) { if (p.C / 4 == 64 * 8) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 8, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 8, 1, false>), g, b, 0, st, p); }
    else if (V == 4) { if (p.C / 4 == 64 * 4) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 4, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 4, 1, false>), g, b, 0, st, p); }
    else if (V == 6) { if (p.C / 4 == 64 * 6) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 6, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 6, 1, false>), g, b, 0, st, p); }
    else if (G == 8) { if (p.C / 4 == 8 * 3) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 8, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 8, 3, 1, false>), g, b, 0, st, p); }
    else if (G == 16) { if (p.C / 4 == 16 * 3) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 16, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 16, 3, 1, false>), g, b, 0, st, p); }
    else if (G == 32) { if (p.C / 4 == 32 * 3) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 32, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 32, 3, 1, false>), g, b, 0, st, p); }
    else { if (p.C / 4 == 64 * 3) hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 3, 1, true>), g, b, 0, st, p); else hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, 64, 3, 1, false>), g, b, 0, st, p); }
    if (p.part) {
        const int slices = grid >= 64 ? 64 : 1;
        if (defer) ln_postpone_reduce(p.part, grid, p.C, p.dw, p.db, slices, st);
        else hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * p.C, 256), slices), dim3(256), 0, st, p.part, grid, p.C, p.dw, p.db);
    }
}

}  // namespace

extern "C" int lnx_layernorm_fwd(const lnx_ln_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->y && a->w && a->b, "lnx_layernorm_fwd: null operand");
    LNX_CHECK(a->M > 0 && a->C > 0 && a->C % 4 == 0 && a->C <= MAXC, "lnx_layernorm_fwd: bad shape M=%d C=%d", a->M, a->C);
    LNX_CHECK(a->ldx % 4 == 0 && a->ldy % 4 == 0, "lnx_layernorm_fwd: ldx/ldy must be multiples of 4");
    LnP p;
    p.x = a->x; p.add = a->add; p.y = a->y; p.w = a->w; p.b = a->b; p.mean = a->mean; p.rstd = a->rstd;
    p.ldx = a->ldx; p.ldy = a->ldy; p.ldadd = a->ldadd;
    p.xmap = RowMap{a->x_map.group, a->x_map.pad, a->x_map.off};
    p.ymap = RowMap{a->y_map.group, a->y_map.pad, a->y_map.off};
    p.M = a->M; p.C = a->C; p.eps = a->eps;
    if (a->y8) {
        LNX_CHECK(a->y8_scales && a->x_dtype == LNX_F32 && a->y_dtype == LNX_BF16 && a->C % 128 == 0 && a->ldy8 % 4 == 0 && (((uintptr_t)a->y8) & 3) == 0 &&
                      a->y_map.group == 0 && a->y_map.pad == 0 && a->y_map.off == 0,
                  "lnx_layernorm_fwd: the MXFP8 output needs x fp32, y bf16, an identity y_map, C %% 128 == 0 and 4-byte aligned rows");
        p.y8 = (unsigned char*)a->y8; p.y8s = (unsigned char*)a->y8_scales; p.ldy8 = a->ldy8;
    }
    hipStream_t st = (hipStream_t)stream;
    const int xi = a->x_dtype, yi = a->y_dtype;
    if (xi == LNX_F32 && yi == LNX_F32) launch_fwd<float, float>(p, st);
    else if (xi == LNX_F32 && yi == LNX_BF16) launch_fwd<float, bf16_t>(p, st);
    else if (xi == LNX_BF16 && yi == LNX_F32) launch_fwd<bf16_t, float>(p, st);
    else if (xi == LNX_BF16 && yi == LNX_BF16) launch_fwd<bf16_t, bf16_t>(p, st);
    else LNX_CHECK(false, "lnx_layernorm_fwd: bad dtypes %d %d", xi, yi);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_layernorm_bwd(const lnx_ln_bwd_args* a, void* stream) {
    LNX_CHECK(a && a->dy && a->x && a->w && a->mean && a->rstd && a->dx, "lnx_layernorm_bwd: null operand");
    LNX_CHECK(a->M > 0 && a->C > 0 && a->C % 4 == 0 && a->C <= MAXC, "lnx_layernorm_bwd: bad shape M=%d C=%d", a->M, a->C);
    LNX_CHECK(a->ldx % 4 == 0 && a->lddy % 4 == 0 && a->lddx % 4 == 0, "lnx_layernorm_bwd: leading dims must be multiples of 4");
    LnBwdP p;
    p.dy = a->dy; p.x = a->x; p.w = a->w; p.mean = a->mean; p.rstd = a->rstd; p.gin = a->gin; p.dx = a->dx; p.dw = a->dw; p.db = a->db;
    p.lddy = a->lddy; p.ldx = a->ldx; p.lddx = a->lddx; p.ldgin = a->ldgin;
    p.dymap = RowMap{a->dy_map.group, a->dy_map.pad, a->dy_map.off};
    p.xmap = RowMap{a->x_map.group, a->x_map.pad, a->x_map.off};
    p.M = a->M; p.C = a->C; p.relu_mask = a->relu_mask;
    p.dx2 = a->dx2; p.rs2 = a->dx2_rowscale; p.lddx2 = a->lddx2; p.rps2 = a->dx2_rows_per_sample > 0 ? a->dx2_rows_per_sample : 1;
    p.dx2_f32 = a->dx2_dtype == LNX_F32 ? 1 : 0;
    if (a->dx2) {
        LNX_CHECK(a->lddx2 % 4 == 0 && (a->dx2_dtype == LNX_F32 || a->dx2_dtype == LNX_BF16), "lnx_layernorm_bwd: bad dx2 layout");
        LNX_CHECK(a->dx2_dtype == LNX_F32 || ((uintptr_t)a->dx2 % 16 == 0 && (a->lddx2 * 2) % 16 == 0), "lnx_layernorm_bwd: dx2 must be 16-byte aligned");
        if (a->dx2_rowscale) LNX_CHECK(a->dx2_rows_per_sample > 0, "lnx_layernorm_bwd: dx2_rowscale needs dx2_rows_per_sample");
    }
    p.dx2_8 = nullptr; p.dx2_8s = nullptr; p.lddx2_8 = 0;
    if (a->dx2_8) {
        LNX_CHECK(a->dx2 && a->dx2_dtype == LNX_BF16 && a->dx2_8_scales && a->x_dtype == LNX_F32 && a->C % 128 == 0 && a->lddx2_8 % 4 == 0 &&
                      (((uintptr_t)a->dx2_8) & 3) == 0,
                  "lnx_layernorm_bwd: the MXFP8 copy of dx2 needs dx2 in bf16, x in fp32, C %% 128 == 0 and 4-byte aligned rows");
        p.dx2_8 = (unsigned char*)a->dx2_8; p.dx2_8s = (unsigned char*)a->dx2_8_scales; p.lddx2_8 = a->lddx2_8;
    }
    p.part = a->ws;
    const int64_t wsf = a->ws ? a->ws_floats : 0;
    hipStream_t st = (hipStream_t)stream;
    const int code = a->dy_dtype * 4 + a->x_dtype * 2 + a->dx_dtype;
    switch (code) {
        case 0: launch_bwd<float, float, float>(p, st, wsf, a->defer != 0); break;
        case 1: launch_bwd<float, float, bf16_t>(p, st, wsf, a->defer != 0); break;
        case 2: launch_bwd<float, bf16_t, float>(p, st, wsf, a->defer != 0); break;
        case 3: launch_bwd<float, bf16_t, bf16_t>(p, st, wsf, a->defer != 0); break;
        case 4: launch_bwd<bf16_t, float, float>(p, st, wsf, a->defer != 0); break;
        case 5: launch_bwd<bf16_t, float, bf16_t>(p, st, wsf, a->defer != 0); break;
        case 6: launch_bwd<bf16_t, bf16_t, float>(p, st, wsf, a->defer != 0); break;
        case 7: launch_bwd<bf16_t, bf16_t, bf16_t>(p, st, wsf, a->defer != 0); break;
        default: LNX_CHECK(false, "lnx_layernorm_bwd: bad dtypes");
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_layernorm_bwd_flush(void* stream) {
    LNX_CHECK(ln_flush((hipStream_t)stream) == 0, "lnx_layernorm_bwd_flush: the postponed calls of this thread were launched on another stream");
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_layernorm_bwd_discard(void) { return ln_discard(); }
