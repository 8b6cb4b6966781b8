// Fused ConvNeXt MLP branch for gfx950 (bf16 storage):
//
//   forward   out = x + rowscale * gamma * ( GELU(ln . W1^T + b1) . W2^T + b2 )
//             (blocks/convnext.py:79-86: pwconv1 -> GELU -> pwconv2 -> LayerScale -> DropPath -> +x)
//   backward  recomputes h = ln . W1^T + b1 per tile, then
//             dz = rowscale*gamma*g ; dgamma += sum rowscale*g*z ; dA = dz . W2 ; dH = dA * GELU'(h) ;
//             dln = dH . W1 ; and writes act = GELU(h), dH, dz for the weight-gradient GEMMs.
//
// Why: at C = 96/192 the two pointwise GEMMs are HBM-bound (arithmetic intensity far below the
// ridge); unfused, the 4C-wide hidden tensor is written twice and read back twice in the forward
// alone.  Here the hidden activation never leaves the chip in the forward, and in the backward it
// is produced once (recompute is ~K=C cheap) instead of saved.
//
// Structure: a 256-thread workgroup owns MT*64 rows (wave = MT m-tiles of 16 rows); the rows'
// ln fragments stay in registers for the whole kernel.  The hidden dimension is walked in chunks
// of 64: the chunk's weight slices arrive by LDS-DMA into a 2-stage ring (prefetch chunk j+1
// under chunk j's MFMAs), the first product is computed TRANSPOSED with a row-slot permutation of
// the weight rows such that its accumulator (after bias+GELU, packed to bf16) is directly the
// B operand of the second product in natural k order -- the hidden tile never touches LDS.
// Weight tiles are laid out [k-step][row][64 B] with the 16-byte unit XOR-swizzled by a 2-bit
// row key (applied on the DMA source address) so every ds_read_b128 group is conflict-free.
#include <stdlib.h>

#include "common.hpp"
#include "../../include/lnx.h"

namespace {

struct CmP {
    const unsigned char* ln;
    const unsigned char* w1;    // [4C, C]
    const unsigned char* w2;    // fwd: [C, 4C]
    const unsigned char* w2t;   // bwd: [4C, C]
    const unsigned char* w1t;   // bwd: [C, 4C]
    const unsigned char* zin;   // bwd: [M, C]
    const float* b1;
    const float* b2;
    const float* gamma;
    const float* rowscale;
    const float* x;
    const float* g;
    float* out;
    unsigned char* z;           // fwd: optional [M, C]
    unsigned char* act;         // bwd out [M, 4C]
    unsigned char* dh;          // bwd out [M, 4C]
    unsigned char* dz;          // bwd out [M, C]
    unsigned char* dln;         // bwd out [M, C]
    float* dgamma;
    int M, C, rps;
    // fused block LayerNorm (LNF forward / LNB backward kernels)
    const unsigned char* y;     // [M, C] bf16 LayerNorm input (the depthwise conv output)
    const float* lnw;           // [C]
    const float* lnb;           // [C]  (forward)
    float eps;                  //      (forward)
    unsigned char* ln_out;      // fwd: optional [M, C] bf16 normalised rows
    float* mean;                // fwd out / bwd in [M]
    float* rstd;
    float* part;                // bwd: per-workgroup partial sums of dw | db, [gridDim.x][2C]
    int64_t part_floats;
    int ln_defer;               // LNB launches: postpone the column-sum fold to lnx_layernorm_bwd_flush (norm.hip)
    int tile_slot;              // resident-weight kernels: tile counter set of this launch (common.hpp), -1 = static stride
    int dz_plain;               // bwd: the dz written to memory is rowscale * g WITHOUT the LayerScale factor (the data gradient inside the
                                // kernel keeps it): operand of a pwconv2 weight gradient that lnx_layerscale_apply_wgrad scales afterwards
    float* dlnw;                // bwd: [C] += LayerNorm weight / bias gradient (host side of the launch: the reduce kernel's targets)
    float* dlnb;
};

__device__ __forceinline__ void mfma16(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// swizzle key of a weight-tile row, table {0, 3, 2, 1}: the four rows of one bank class (rows 4 apart
// share the same four 16-byte bank slots) get distinct slots in every ds_read_b128 lane group
__device__ __forceinline__ int key4(int a) { return (4 - (a & 3)) & 3; }

#define CM_DS_READ128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr) : "memory")

// bytes of one weight part (W1 slice [NK][64][64B] == W2 slice [2][C][64B]) for NK = C/32
template <int NK> struct Geo {
    static constexpr int C = 32 * NK;
    static constexpr int PART = NK * 4096;
    static constexpr int CT = C / 16;           // c tiles
};

// issue the DMA of one "n-major" part: rows = 64 hidden units (n0..n0+63), columns = C channels;
// source matrix [4C, C] row-major.  LDS image [ks][row][4 units]; unit u' holds source unit u' ^ key4(row>>3)
template <int NK, int NW>
__device__ __forceinline__ void dma_nmajor(unsigned char* lds_part, const unsigned char* W, int n0, int wave, int lane) {
    constexpr int C = Geo<NK>::C;
    constexpr int NINS = 4 * NK;  // 1 KiB instructions, spread over the NW waves
    static_assert(NINS % NW == 0, "DMA pieces must divide evenly over the waves");
#pragma unroll
    for (int t = 0; t < NINS / NW; ++t) {
        const int i = wave + NW * t;
        const int ks = i >> 2, rb = i & 3;
        const int row = 16 * rb + (lane >> 2), u = lane & 3;
        const unsigned char* src = W + ((int64_t)(n0 + row) * C + ks * 32 + ((u ^ key4(row >> 3)) << 3)) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_part + i * 1024), 16, 0, 0);
    }
}

// issue the DMA of one "c-major" part: rows = C channels, columns = 64 hidden units (n0..n0+63);
// source matrix [C, 4C] row-major.  LDS image [ks2][c][4 units]; unit u' holds source unit u' ^ key4(c>>2)
template <int NK, int NW>
__device__ __forceinline__ void dma_cmajor(unsigned char* lds_part, const unsigned char* W, int n0, int wave, int lane) {
    constexpr int C = Geo<NK>::C;
    constexpr int RB = C / 16;
    constexpr int NINS = 2 * RB;  // == 4 NK
#pragma unroll
    for (int t = 0; t < NINS / NW; ++t) {
        const int i = wave + NW * t;
        const int ks2 = i / RB, rb = i % RB;
        const int row = 16 * rb + (lane >> 2), u = lane & 3;
        const unsigned char* src = W + ((int64_t)row * (4 * C) + n0 + ks2 * 32 + ((u ^ key4(row >> 2)) << 3)) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_part + i * 1024), 16, 0, 0);
    }
}

// first-kind product:  H^T[n][m] = sum_k Wn[n][k] * X[m][k]   (n: 64 permuted rows of an n-major part)
template <int NK, int MT>
__device__ __forceinline__ void prod_nmajor(f32x4_t (&h)[4][MT], uint32_t part_addr, int s, int g, const uint4 (&xf)[MT][NK]) {
    // fragment row of tile nt for lane s: 32(nt>>1) + 8(s>>2) + 4(nt&1) + (s&3); unit g ^ key4(row>>3) = g ^ key4(s>>2)
    const uint32_t unit = (uint32_t)((g ^ key4(s >> 2)) << 4);
    const uint32_t base = part_addr + (uint32_t)((8 * (s >> 2) + (s & 3)) * 64) + unit;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) h[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        uint4 wf[4];
        const uint32_t a = base + ks * 4096;
        CM_DS_READ128(wf[0], a, 0);      // nt = 0: row offset 0
        CM_DS_READ128(wf[1], a, 256);    // nt = 1: +4 rows
        CM_DS_READ128(wf[2], a, 2048);   // nt = 2: +32 rows
        CM_DS_READ128(wf[3], a, 2304);   // nt = 3: +36 rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mfma16(h[nt][mt], wf[nt], xf[mt][ks]);
    }
}

// the same with KB k-steps of fragment reads per wait (4 KB registers): the resident kernels, whose waves run without barriers and
// in pairs of the same shape per SIMD, expose every LDS round trip they wait for
template <int NK, int MT, int KB>
__device__ __forceinline__ void prod_nmajor_b(f32x4_t (&h)[4][MT], uint32_t part_addr, int s, int g, const uint4 (&xf)[MT][NK]) {
    const uint32_t unit = (uint32_t)((g ^ key4(s >> 2)) << 4);
    const uint32_t base = part_addr + (uint32_t)((8 * (s >> 2) + (s & 3)) * 64) + unit;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) h[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < NK; k0 += KB) {
        uint4 wf[KB][4];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (k0 + kb < NK) {
                const uint32_t a = base + (k0 + kb) * 4096;
                CM_DS_READ128(wf[kb][0], a, 0);
                CM_DS_READ128(wf[kb][1], a, 256);
                CM_DS_READ128(wf[kb][2], a, 2048);
                CM_DS_READ128(wf[kb][3], a, 2304);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (k0 + kb < NK) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mfma16(h[nt][mt], wf[kb][nt], xf[mt][k0 + kb]);
            }
        }
    }
}

// second-kind product:  O^T[c][m] += sum_n Wc[c][n] * P[n][m]   (n over the chunk's 64 hidden units)
template <int NK, int MT>
__device__ __forceinline__ void prod_cmajor(f32x4_t (&o)[Geo<NK>::CT][MT], uint32_t part_addr, int s, int g, const uint4 (&pf)[MT][2]) {
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    // row c = 16 ct + s ; unit g ^ key4(c>>2) = g ^ key4(s>>2)  (16 ct does not change (c>>2)&3)
    const uint32_t base = part_addr + (uint32_t)(s * 64) + (uint32_t)((g ^ key4(s >> 2)) << 4);
    constexpr int GB = CT % 6 == 0 ? 6 : (CT % 4 == 0 ? 4 : CT);  // fragment reads per batch (one wait per batch)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
#pragma unroll
        for (int c0 = 0; c0 < CT; c0 += GB) {
            uint4 wf[GB];
#pragma unroll
            for (int b = 0; b < GB; ++b) {
                const uint32_t a = base + (uint32_t)(ks2 * C * 64 + (c0 + b) * 1024);
                CM_DS_READ128(wf[b], a, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < GB; ++b)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) mfma16(o[c0 + b][mt], wf[b], pf[mt][ks2]);
        }
    }
}

// Software-pipelined forms of the two products for the kernels that have registers to spare: the fragment reads of
// batch b+1 are in flight while the MFMAs of batch b issue (LDS returns in order, so a counted lgkmcnt wait is enough).
// In the plain forms every batch of 4-6 MFMAs (64-96 cycles) waits out a full LDS round trip first.
template <int NK, int MT>
__device__ __forceinline__ void prod_nmajor_pipe(f32x4_t (&h)[4][MT], uint32_t part_addr, int s, int g, const uint4 (&xf)[MT][NK]) {
    const uint32_t unit = (uint32_t)((g ^ key4(s >> 2)) << 4);
    const uint32_t base = part_addr + (uint32_t)((8 * (s >> 2) + (s & 3)) * 64) + unit;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) h[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 wf[2][4];
    CM_DS_READ128(wf[0][0], base, 0);
    CM_DS_READ128(wf[0][1], base, 256);
    CM_DS_READ128(wf[0][2], base, 2048);
    CM_DS_READ128(wf[0][3], base, 2304);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        if (ks + 1 < NK) {
            const uint32_t a = base + (ks + 1) * 4096;
            CM_DS_READ128(wf[(ks + 1) & 1][0], a, 0);
            CM_DS_READ128(wf[(ks + 1) & 1][1], a, 256);
            CM_DS_READ128(wf[(ks + 1) & 1][2], a, 2048);
            CM_DS_READ128(wf[(ks + 1) & 1][3], a, 2304);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mfma16(h[nt][mt], wf[ks & 1][nt], xf[mt][ks]);
    }
}

template <int NK, int MT>
__device__ __forceinline__ void prod_cmajor_pipe(f32x4_t (&o)[Geo<NK>::CT][MT], uint32_t part_addr, int s, int g, const uint4 (&pf)[MT][2]) {
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    const uint32_t base = part_addr + (uint32_t)(s * 64) + (uint32_t)((g ^ key4(s >> 2)) << 4);
    constexpr int GB = 4;                 // fragment reads per batch
    constexpr int NB = 2 * CT / GB;       // batches over (ks2, ct)
    static_assert(CT % GB == 0, "channel tiles must split into batches of 4");
    uint4 wf[2][GB];
#pragma unroll
    for (int b = 0; b < GB; ++b) CM_DS_READ128(wf[0][b], base + (uint32_t)(b * 1024), 0);
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
        const int ks2 = (bt * GB) / CT, c0 = (bt * GB) % CT;
        if (bt + 1 < NB) {
            const int ks2n = ((bt + 1) * GB) / CT, c0n = ((bt + 1) * GB) % CT;
#pragma unroll
            for (int b = 0; b < GB; ++b) CM_DS_READ128(wf[(bt + 1) & 1][b], base + (uint32_t)(ks2n * C * 64 + (c0n + b) * 1024), 0);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < GB; ++b)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mfma16(o[c0 + b][mt], wf[bt & 1][b], pf[mt][ks2]);
    }
}

__device__ __forceinline__ uint4 pack8(const f32x4_t& lo, const f32x4_t& hi) {
    Vec16<bf16_t> v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v.set(j, lo[j]);
        v.set(4 + j, hi[j]);
    }
    return v.raw;
}

// ------------------------------------------------------------------------------------
// Block LayerNorm (blocks/convnext.py:77 `self.norm`, eps 1e-6) inside the conv-MLP kernels.  A wave owns whole rows in both
// of its layouts -- the A-fragment layout (lane (s, g): row s, channels 32 ks + 8 g .. + 7) and the accumulator layout (row s,
// channels 16 ct + 4 g .. + 3) -- so a row statistic is a per-lane sum plus two cross-lane steps over g, and the separate
// LayerNorm passes (forward: read y, write ln; backward: read dln and y, write dy) disappear into kernels that are HBM-bound.
// Same arithmetic as norm.hip: two-pass mean / centred variance, 1 / sqrtf, fma(v * rstd, w, b).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float g_sum(float v) {  // over the four lanes (g) of a row
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes s of a DPP row (every lane gets it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return v;
}
// forward: the raw y fragments of one row -> its normalised (bf16) fragments; w / b: [C] (global or LDS)
template <int NK>
__device__ __forceinline__ void ln_row_frags(uint4 (&xf)[NK], const float* w, const float* b, int g, float eps, float& mu, float& rs) {
    constexpr float invC = 1.0f / (float)Geo<NK>::C;
    float v[NK][8];
    float sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        Vec16<bf16_t> t;
        t.raw = xf[ks];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[ks][j] = t.get(j);
        sum += ((v[ks][0] + v[ks][1]) + (v[ks][2] + v[ks][3])) + ((v[ks][4] + v[ks][5]) + (v[ks][6] + v[ks][7]));
    }
    mu = g_sum(sum) * invC;
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[ks][j] -= mu;
            sq = fmaf(v[ks][j], v[ks][j], sq);
        }
    rs = 1.0f / sqrtf(fmaf(g_sum(sq), invC, eps));
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int c = ks * 32 + 8 * g;
        const float4 w0 = *reinterpret_cast<const float4*>(w + c), w1 = *reinterpret_cast<const float4*>(w + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        Vec16<bf16_t> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.set(j, fmaf(v[ks][j] * rs, wv[j], bv[j]));
        xf[ks] = o.raw;
    }
}
// backward: dln (this wave's accumulators of m-tile MTI, rounded to bf16 as the unfused path stores them) -> dy of row m into
// p.dln, column sums of dln * xhat and dln into the workgroup's LDS partials `dls` ([2C]).  EVERY lane must call (cross-lane sums);
// lnws: [C] LayerNorm weight in LDS.
// The row's y (accumulator layout) and statistics are requested by ln_bwd_fetch at the START of the tile, with the tile's other
// operands: fetched here, at the end, every tile would pay one exposed memory round trip more (measured: +15 % kernel time).
template <int NK>
struct LnRow {
    uint2 yr[Geo<NK>::CT];
    float mu, rs;
};
template <int NK>
__device__ __forceinline__ void ln_bwd_fetch(LnRow<NK>& r, const CmP& p, int m, int g) {
    constexpr int C = Geo<NK>::C;
    const int mc = m < p.M ? m : p.M - 1;
#pragma unroll
    for (int ct = 0; ct < Geo<NK>::CT; ++ct) r.yr[ct] = *reinterpret_cast<const uint2*>(p.y + ((int64_t)mc * C + ct * 16 + 4 * g) * 2);
    r.mu = p.mean[mc];
    r.rs = p.rstd[mc];
}
template <int NK, int MT, int MTI>
__device__ __forceinline__ void ln_bwd_rows(const CmP& p, const f32x4_t (&dl)[Geo<NK>::CT][MT], const LnRow<NK>& row, int m, int s, int g, const float* lnws,
                                            float* dls) {
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    constexpr float invC = 1.0f / (float)C;
    const bool mv = m < p.M;        // a lane beyond M holds row M - 1 again (clamped loads): it stores the same dy as its live twin
    const int mc = mv ? m : p.M - 1;  // and contributes nothing to the column sums
    const uint2 (&yr)[CT] = row.yr;
    const float mu = row.mu, rs = row.rs;
    float xh[CT][4], gv[CT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + 4 * g;
        const float4 w4 = *reinterpret_cast<const float4*>(lnws + c);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
        const bf16_t* yh = reinterpret_cast<const bf16_t*>(&yr[ct]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = (float)(bf16_t)dl[ct][MTI][r];
            const float x = ((float)yh[r] - mu) * rs;
            xh[ct][r] = x;
            gv[ct][r] = d * wv[r];
            s1 += gv[ct][r];
            s2 = fmaf(gv[ct][r], x, s2);
            const float dm = mv ? d : 0.f;
            const float a = row16_sum(dm * x), b = row16_sum(dm);
            if (s == 0) {
                atomicAdd(&dls[c + r], a);
                atomicAdd(&dls[C + c + r], b);
            }
        }
    }
    const float m1 = g_sum(s1) * invC, m2 = g_sum(s2) * invC;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        uint2 v;
        bf16_t* vh = reinterpret_cast<bf16_t*>(&v);
#pragma unroll
        for (int r = 0; r < 4; ++r) vh[r] = (bf16_t)(rs * (gv[ct][r] - m1 - xh[ct][r] * m2));
        *reinterpret_cast<uint2*>(p.dln + ((int64_t)mc * C + ct * 16 + 4 * g) * 2) = v;
    }
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
// Diagnostic build only (-DCM_STAMP, tools/build_stamp.sh): per-phase s_memtime sums of wave 0 of one workgroup
#ifdef CM_STAMP
__device__ unsigned long long g_cm_stamp[8];
#define CM_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define CM_T(i) do { } while (0)
#endif

// LNF: the kernel reads the LayerNorm INPUT p.y and normalises it in registers (ln_row_frags); p.ln_out / p.mean / p.rstd (optional)
// receive what the backward needs
template <int NK, int MT, int NW, bool LNF>
__global__ __launch_bounds__(64 * NW) void convmlp_fwd_kernel(const CmP p) {
#ifdef CM_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    constexpr int PART = Geo<NK>::PART;
    constexpr int STAGE = 2 * PART;
    constexpr int NCH = 4 * C / 64;
    constexpr int NLD = 8 * NK / NW;  // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3 stages][W1 part | W2 part] + b1 [4C floats]
    float* b1s = reinterpret_cast<float*>(smem + 3 * STAGE);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int m_base = blockIdx.x * (16 * NW * MT) + wave * (16 * MT);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    for (int i = threadIdx.x; i < 4 * C; i += 64 * NW) b1s[i] = p.b1[i];

    // this lane's ln fragments: row m (clamped), channels ks*32 + 8g .. +7
    uint4 xf[MT][NK];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m_base + mt * 16 + s;
        if (m >= p.M) m = p.M - 1;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) xf[mt][ks] = ld16((LNF ? p.y : p.ln) + ((int64_t)m * C + ks * 32 + 8 * g) * 2);
    }
    if constexpr (LNF) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float mu, rs;
            ln_row_frags<NK>(xf[mt], p.lnw, p.lnb, g, p.eps, mu, rs);
            const int m = m_base + mt * 16 + s;
            if (m < p.M) {
                if (p.ln_out) {
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) st16(p.ln_out + ((int64_t)m * C + ks * 32 + 8 * g) * 2, xf[mt][ks]);
                }
                if (p.mean && g == 0) {
                    p.mean[m] = mu;
                    p.rstd[m] = rs;
                }
            }
        }
    }
    f32x4_t o[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o[ct][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // b1s staged (plain LDS stores) before any LDS-DMA is in flight
    // 3-stage ring, two chunks of weights in flight: one chunk's MFMA + GELU work (~1100 cycles) does not cover the
    // ~2000+ cycle LDS-DMA latency, so the double-buffered form stalled on every chunk.  One barrier per chunk: it
    // publishes chunk j (every wave has waited for its own pieces) and retires the reads of chunk j-1, whose slot
    // chunk j+2 is then issued into.
    dma_nmajor<NK, NW>(smem, p.w1, 0, wave, lane);
    dma_cmajor<NK, NW>(smem + PART, p.w2, 0, wave, lane);
    dma_nmajor<NK, NW>(smem + STAGE, p.w1, 64, wave, lane);
    dma_cmajor<NK, NW>(smem + STAGE + PART, p.w2, 64, wave, lane);
    CM_T(0);
    for (int j = 0; j < NCH; ++j) {
        const int stg = j % 3;
        if (j + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CM_T(1);
        __builtin_amdgcn_s_barrier();
        CM_T(2);
        if (j + 2 < NCH) {
            unsigned char* nx = smem + ((j + 2) % 3) * STAGE;
            dma_nmajor<NK, NW>(nx, p.w1, 64 * (j + 2), wave, lane);
            dma_cmajor<NK, NW>(nx + PART, p.w2, 64 * (j + 2), wave, lane);
        }
        const uint32_t st = lds0 + stg * STAGE;
        f32x4_t h[4][MT];
        CM_T(3);
        prod_nmajor_pipe<NK, MT>(h, st, s, g, xf);
        CM_T(4);
        // bias + exact-erf GELU on the accumulator; row 4g+r of tile nt <-> hidden n = 64j + 32(nt>>1) + 8g + 4(nt&1) + r
        uint4 pf[MT][2];
        {
            f32x4_t bv[4];
            const uint32_t ba = lds0 + 3 * STAGE + (uint32_t)((64 * j + 8 * g) * 4);
            CM_DS_READ128(bv[0], ba, 0);
            CM_DS_READ128(bv[1], ba, 16);
            CM_DS_READ128(bv[2], ba, 128);
            CM_DS_READ128(bv[3], ba, 144);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 4; n2 += 2) {  // four element pairs in lockstep (common.hpp "_n" forms)
                    f32x4_t hv[2] = {h[n2][mt], h[n2 + 1][mt]};
                    gelu4_bias_n<2>(hv, &bv[n2]);
                    h[n2][mt] = hv[0];
                    h[n2 + 1][mt] = hv[1];
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                pf[mt][0] = pack8(h[0][mt], h[1][mt]);
                pf[mt][1] = pack8(h[2][mt], h[3][mt]);
            }
        }
        CM_T(5);
        prod_cmajor_pipe<NK, MT>(o, st + PART, s, g, pf);
        CM_T(6);
    }
    // epilogue: lane (s = row m, g) holds channels c = 16 ct + 4 g + r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m_base + mt * 16 + s;
        if (m >= p.M) continue;
        const float rs = p.rowscale ? p.rowscale[m / p.rps] : 1.0f;
        // all loads first: as far as the compiler knows the stores below may alias them, and interleaving
        // would serialise one full memory round trip per channel tile
        float4 xrv[CT], b2v[CT], gmv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + 4 * g;
            xrv[ct] = *reinterpret_cast<const float4*>(p.x + (int64_t)m * C + c);
            b2v[ct] = *reinterpret_cast<const float4*>(p.b2 + c);
            gmv[ct] = *reinterpret_cast<const float4*>(p.gamma + c);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + 4 * g;
            const float4 b2 = b2v[ct], gm = gmv[ct], xr = xrv[ct];
            const float z0 = o[ct][mt][0] + b2.x, z1 = o[ct][mt][1] + b2.y, z2 = o[ct][mt][2] + b2.z, z3 = o[ct][mt][3] + b2.w;
            if (p.z) {
                uint2 zz;
                bf16_t* zh = reinterpret_cast<bf16_t*>(&zz);
                zh[0] = (bf16_t)z0; zh[1] = (bf16_t)z1; zh[2] = (bf16_t)z2; zh[3] = (bf16_t)z3;
                *reinterpret_cast<uint2*>(p.z + ((int64_t)m * C + c) * 2) = zz;
            }
            *reinterpret_cast<float4*>(p.out + (int64_t)m * C + c) =
                make_float4(xr.x + rs * gm.x * z0, xr.y + rs * gm.y * z1, xr.z + rs * gm.z * z2, xr.w + rs * gm.w * z3);
        }
    }
    CM_T(7);
#ifdef CM_STAMP
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_cm_stamp[i] = tsum[i];
#endif
}
#ifdef CM_STAMP
extern "C" int lnx_dbg_convmlp_stamps(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cm_stamp), 64); }
#endif

// ------------------------------------------------------------------------------------
// backward (data side): act, dH, dz, dln, dgamma
// ------------------------------------------------------------------------------------
// ST: also store act = GELU(h) and dH ([M, 4C] each), the operands of the two weight-gradient GEMMs the plan launches after
// this kernel (ST = false: a caller that only wants the data gradient)
// LNB: the LayerNorm backward runs in the epilogue (ln_bwd_rows): p.dln receives the gradient wrt the LayerNorm INPUT p.y, and
// the workgroup's column sums for the LayerNorm weight / bias gradient go to p.part (summed by ln_partials_reduce_kernel)
template <int NK, int MT, int NW, bool ST, bool LNB, bool DG>
__global__ __launch_bounds__(64 * NW) void convmlp_bwd_kernel(const CmP p) {
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    constexpr int PART = Geo<NK>::PART;
    constexpr int STAGE = 3 * PART;  // W1 slice | W2^T slice | W1^T slice
    constexpr int NCH = 4 * C / 64;
    constexpr int NLD = 12 * NK / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* b1s = reinterpret_cast<float*>(smem + 2 * STAGE);
    float* dgs = b1s + 4 * C;  // [C] per-workgroup dgamma partial
    float* lws = dgs + C;      // LNB: [C] LayerNorm weight, [2C] column-sum partials
    float* dls = lws + C;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;
    const int m_base = blockIdx.x * (16 * NW * MT) + wave * (16 * MT);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    for (int i = threadIdx.x; i < 4 * C; i += 64 * NW) b1s[i] = p.b1[i];
    for (int i = threadIdx.x; i < C; i += 64 * NW) {
        dgs[i] = 0.f;
        if constexpr (LNB) {
            lws[i] = p.lnw[i];
            dls[i] = dls[C + i] = 0.f;
        }
    }
    __syncthreads();

    uint4 xf[MT][NK], zf[MT][NK];  // ln fragments; dz = rowscale*gamma*g fragments (bf16)
    bool mvalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = m_base + mt * 16 + s;
        mvalid[mt] = m < p.M;
        if (!mvalid[mt]) m = p.M - 1;  // such a lane redoes row M - 1 and stores the same bytes as its live twin; only the column sums mask it
        const float rs = p.rowscale ? p.rowscale[m / p.rps] : 1.0f;
        // loads of every k-step first, then compute + dz stores (see the forward epilogue note)
        float4 g0v[NK], g1v[NK], a0v[NK], a1v[NK];
        uint4 zraw[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = ks * 32 + 8 * g;
            xf[mt][ks] = ld16(p.ln + ((int64_t)m * C + c) * 2);
            g0v[ks] = *reinterpret_cast<const float4*>(p.g + (int64_t)m * C + c);
            g1v[ks] = *reinterpret_cast<const float4*>(p.g + (int64_t)m * C + c + 4);
            a0v[ks] = *reinterpret_cast<const float4*>(p.gamma + c);
            a1v[ks] = *reinterpret_cast<const float4*>(p.gamma + c + 4);
            if constexpr (DG) zraw[ks] = ld16(p.zin + ((int64_t)m * C + c) * 2);
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int c = ks * 32 + 8 * g;
            const float4 g0 = g0v[ks], g1 = g1v[ks], a0 = a0v[ks], a1 = a1v[ks];
            Vec16<bf16_t> zin, dzv, dzs;
            if constexpr (DG) zin.raw = zraw[ks];
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sg = rs * gv[j];
                dzv.set(j, sg * av[j]);
                dzs.set(j, p.dz_plain ? sg : sg * av[j]);  // what goes to memory for the weight-gradient product (see CmP::dz_plain)
                if constexpr (DG) {
                    // dgamma partial: this lane's 8 channels of its row; reduced over rows below
                    const float dgp = mvalid[mt] ? sg * zin.get(j) : 0.f;
                    // sum over the 16 rows (lanes s) of this m-tile that share g: xor-shuffle over the low 4 lane bits
                    float t = dgp;
                    t += __shfl_xor(t, 1, 64);
                    t += __shfl_xor(t, 2, 64);
                    t += __shfl_xor(t, 4, 64);
                    t += __shfl_xor(t, 8, 64);
                    if (s == 0) atomicAdd(&dgs[c + j], t);  // LDS atomic, 4 lanes per wave
                }
            }
            zf[mt][ks] = dzv.raw;
            st16(p.dz + ((int64_t)m * C + c) * 2, dzs.raw);
        }
    }
    f32x4_t dl[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dl[ct][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    LnRow<NK> lrow;
    if constexpr (LNB) ln_bwd_fetch<NK>(lrow, p, m_base + s, g);

    __syncthreads();
    dma_nmajor<NK, NW>(smem, p.w1, 0, wave, lane);
    dma_nmajor<NK, NW>(smem + PART, p.w2t, 0, wave, lane);
    dma_cmajor<NK, NW>(smem + 2 * PART, p.w1t, 0, wave, lane);
    for (int j = 0; j < NCH; ++j) {
        const int stg = j & 1;
        __builtin_amdgcn_s_barrier();
        if (j + 1 < NCH) {
            unsigned char* nx = smem + (stg ^ 1) * STAGE;
            dma_nmajor<NK, NW>(nx, p.w1, 64 * (j + 1), wave, lane);
            dma_nmajor<NK, NW>(nx + PART, p.w2t, 64 * (j + 1), wave, lane);
            dma_cmajor<NK, NW>(nx + 2 * PART, p.w1t, 64 * (j + 1), wave, lane);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");  // (allowing the previous chunk's act / dH stores to stay in flight here, vmcnt(NLD + 4 MT), changes nothing: measured)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const uint32_t st = lds0 + stg * STAGE;
        f32x4_t h[4][MT], da[4][MT];
        prod_nmajor_pipe<NK, MT>(h, st, s, g, xf);          // h  = ln . W1^T   (pre-bias)
        prod_nmajor_pipe<NK, MT>(da, st + PART, s, g, zf);  // dA = dz . W2
        f32x4_t bv[4];
        const uint32_t ba = lds0 + 2 * STAGE + (uint32_t)((64 * j + 8 * g) * 4);
        CM_DS_READ128(bv[0], ba, 0);
        CM_DS_READ128(bv[1], ba, 16);
        CM_DS_READ128(bv[2], ba, 128);
        CM_DS_READ128(bv[3], ba, 144);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n2 = 0; n2 < 4; n2 += 2) {  // h = act = GELU(h + b), dH = dA * GELU'(h + b); four element pairs in lockstep
                f32x4_t hv[2] = {h[n2][mt], h[n2 + 1][mt]}, dv[2] = {da[n2][mt], da[n2 + 1][mt]};
                gelu4_bias_grad_n<2>(hv, &bv[n2], dv);
                h[n2][mt] = hv[0];
                h[n2 + 1][mt] = hv[1];
                da[n2][mt] = dv[0];
                da[n2 + 1][mt] = dv[1];
            }
        uint4 pa[MT][2], pd[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            pa[mt][0] = pack8(h[0][mt], h[1][mt]);
            pa[mt][1] = pack8(h[2][mt], h[3][mt]);
            pd[mt][0] = pack8(da[0][mt], da[1][mt]);
            pd[mt][1] = pack8(da[2][mt], da[3][mt]);
            // packed element j of k-step ks2 <-> hidden n = 64 j_chunk + 32 ks2 + 8 g + j : 16 contiguous bytes per row
            const int m = min(m_base + mt * 16 + s, p.M - 1);
            if constexpr (ST) {
                const int64_t off = ((int64_t)m * (4 * C) + 64 * j + 8 * g) * 2;
                st16(p.act + off, pa[mt][0]);
                st16(p.act + off + 64, pa[mt][1]);
                st16(p.dh + off, pd[mt][0]);
                st16(p.dh + off + 64, pd[mt][1]);
            }
        }
        prod_cmajor_pipe<NK, MT>(dl, st + 2 * PART, s, g, pd);  // dln += dH . W1
    }
    if constexpr (LNB) {
        static_assert(MT == 1, "ln_bwd_rows is instantiated per m-tile");
        ln_bwd_rows<NK, MT, 0>(p, dl, lrow, m_base + s, s, g, lws, dls);
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(m_base + mt * 16 + s, p.M - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + 4 * g;
                uint2 v;
                bf16_t* vh = reinterpret_cast<bf16_t*>(&v);
                vh[0] = (bf16_t)dl[ct][mt][0]; vh[1] = (bf16_t)dl[ct][mt][1]; vh[2] = (bf16_t)dl[ct][mt][2]; vh[3] = (bf16_t)dl[ct][mt][3];
                *reinterpret_cast<uint2*>(p.dln + ((int64_t)m * C + c) * 2) = v;
            }
        }
    }
    __syncthreads();
    if constexpr (DG)
        for (int i = threadIdx.x; i < C; i += 64 * NW) atomicAdd(p.dgamma + i, dgs[i]);
    if constexpr (LNB)
        for (int i = threadIdx.x; i < 2 * C; i += 64 * NW) p.part[(int64_t)blockIdx.x * (2 * C) + i] = dls[i];
}


// ------------------------------------------------------------------------------------
// Weight-resident persistent variants for C <= 96 (the whole W1 and W2 / W2^T fit in LDS:
// 2 * 4C * C * 2 B = 144 KiB at C = 96).  One 512-thread workgroup per CU loads the weights once,
// then its 8 waves stream 32-row tiles independently with NO barrier in the main loop, so the two
// waves of a SIMD drift apart and one wave's GELU VALU work runs under the other's MFMAs.
// ------------------------------------------------------------------------------------
template <int NK>
__device__ __forceinline__ void dma_all_nmajor(unsigned char* img, const unsigned char* W, int wave, int lane) {
    constexpr int C = Geo<NK>::C;
    constexpr int NCH = 4 * C / 64;
    constexpr int NINS = 4 * NK;
    for (int q = wave; q < NCH * NINS; q += 8) {
        const int j = q / NINS, i = q % NINS;
        const int ks = i >> 2, rb = i & 3;
        const int row = 16 * rb + (lane >> 2), u = lane & 3;
        const unsigned char* src = W + ((int64_t)(64 * j + row) * C + ks * 32 + ((u ^ key4(row >> 3)) << 3)) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(img + j * Geo<NK>::PART + i * 1024), 16, 0, 0);
    }
}
template <int NK>
__device__ __forceinline__ void dma_all_cmajor(unsigned char* img, const unsigned char* W, int wave, int lane) {
    constexpr int C = Geo<NK>::C;
    constexpr int NCH = 4 * C / 64;
    constexpr int RB = C / 16;
    constexpr int NINS = 2 * RB;
    for (int q = wave; q < NCH * NINS; q += 8) {
        const int j = q / NINS, i = q % NINS;
        const int ks2 = i / RB, rb = i % RB;
        const int row = 16 * rb + (lane >> 2), u = lane & 3;
        const unsigned char* src = W + ((int64_t)row * (4 * C) + 64 * j + ks2 * 32 + ((u ^ key4(row >> 2)) << 3)) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(img + j * Geo<NK>::PART + i * 1024), 16, 0, 0);
    }
}

typedef __attribute__((ext_vector_type(4))) short cm_s16x4;
// O^T[c][m] += sum_n Wn[n][c] * P[n][m] with Wn an n-major part (rows = hidden n, cols = channels c):
// the A operand (rows c, k = n) is gathered with transposed reads from the n-major image
template <int NK, int MT>
__device__ __forceinline__ void prod_tr_nmajor(f32x4_t (&o)[Geo<NK>::CT][MT], uint32_t part_addr, int s, int g, const uint4 (&pf)[MT][2]) {
    constexpr int CT = Geo<NK>::CT;
    const int q = s >> 2, pp = s & 3;
    // One batch of transposed reads per k-step (all CT channel tiles: 4 CT registers, free at this point of the chunk -- the hidden
    // accumulators are packed), ONE wait, then the MFMAs.  Per channel tile (read, wait, MFMA) exposed 12 LDS round trips per hidden
    // chunk of the backward: with two waves per SIMD of the same shape that is idle time, not overlap.
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
        const int n = 32 * ks2 + 8 * g + q;  // (n >> 3) & 3 == g for this row and the one 4 below
        uint2 lo[CT], hi[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int ks = ct >> 1;
            const int unit = 2 * (ct & 1) + (pp >> 1);
            const uint32_t a = part_addr + (uint32_t)(ks * 4096 + n * 64 + ((unit ^ key4(g)) << 4) + (pp & 1) * 8);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[ct]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(hi[ct]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const uint4 wf = make_uint4(lo[ct].x, lo[ct].y, hi[ct].x, hi[ct].y);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) mfma16(o[ct][mt], wf, pf[mt][ks2]);
        }
    }
}

template <bool V> struct BoolC { static constexpr bool value = V; };
// SAVE: the outputs a backward needs are written: 1 = z and, with LNF, the normalised rows + row statistics; 2 (round 4, LNF) = the
// LayerNorm outputs without z -- the LayerScale gradient then comes from the pwconv2 weight gradient (lnx_layerscale_apply_wgrad)
template <int NK, bool LNF, int SAVE>
__global__ __launch_bounds__(512) void convmlp_fwd_res_kernel(const CmP p) {
    constexpr int MT = 2;
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    constexpr int PART = Geo<NK>::PART;
    constexpr int NCH = 4 * C / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [W1 image NCH*PART][W2 image NCH*PART][b1 4C floats]
    unsigned char* w1img = smem;
    unsigned char* w2img = smem + NCH * PART;
    float* b1s = reinterpret_cast<float*>(smem + 2 * NCH * PART);
    float* b2s = b1s + 4 * C;
    float* gms = b2s + C;
    float* lws = gms + C;  // LNF: LayerNorm weight | bias
    float* lbs = lws + C;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    for (int i = threadIdx.x; i < 4 * C; i += 512) b1s[i] = p.b1[i];
    for (int i = threadIdx.x; i < C; i += 512) {
        b2s[i] = p.b2[i];
        gms[i] = p.gamma[i];
        if constexpr (LNF) {
            lws[i] = p.lnw[i];
            lbs[i] = p.lnb[i];
        }
    }
    // tile scheduling (common.hpp): each WAVE draws its 32-row tiles from the launch's counter, the first one here, under the
    // weight DMA (its round trip is covered by the wait below); static stride when tile_slot < 0
    const bool dyn = p.tile_slot >= 0;
    const TileShare sh = tile_share((p.M + 31) / 32);
    unsigned* const ctr = &g_tile_ctr[dyn ? p.tile_slot : 0][sh.part][0];
    int drawn = dyn ? sched_draw_counted(ctr) : 0;
    dma_all_nmajor<NK>(w1img, p.w1, wave, lane);
    dma_all_cmajor<NK>(w2img, p.w2, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Software pipeline over this wave's tiles: the ln fragments of tile t+1 and the residual rows of
    // tile t are requested BEFORE tile t's MFMA/GELU work, so every HBM round trip runs under
    // arithmetic (all waves of the chip would otherwise alternate in lockstep between a compute
    // phase and a memory phase, which is what the ablation showed: the phase times simply added up).
    // No memory operation of the tile loop sits under a branch: with one in-order counter for loads and stores the compiler can only
    // count ("wait until the loads issued before these N stores have landed") when every path issues the same operations -- under a
    // branch it drains the counter instead, which made every tile wait for the previous tile's store acknowledgements AND the next
    // tile's prefetch (round 3).  So: rows beyond M are clamped to M - 1 (such a lane recomputes row M - 1 and stores the same bytes
    // as its live twin), the optional outputs are a template parameter (SAVE), the DropPath scale is loaded unconditionally.
    const int ntile = (p.M + 31) / 32;
    auto load_x = [&](uint4 (&dst)[MT][NK], int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(tile * 32 + mt * 16 + s, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) dst[mt][ks] = ld16((LNF ? p.y : p.ln) + ((int64_t)m * C + ks * 32 + 8 * g) * 2);
        }
    };
    // this XCD's tiles [xbase, xbase + xcnt) and its waves; positions are drawn (or strided over, LNX_TILE_SCHED=static) within them
    const int xbase = sh.base, xcnt = sh.cnt, xwaves = 8 * sh.workers;
    const int last_draw = xcnt + xwaves - 1;  // one draw per tile + one failed draw per wave: the counter's last answer
    int pos = dyn ? __builtin_amdgcn_readfirstlane(drawn) : sh.index * 8 + wave;
    bool reset_ctr = dyn && pos == last_draw;
    uint4 xf[MT][NK];
    if (pos < xcnt) load_x(xf, xbase + pos);
    const float* rsp = p.rowscale ? p.rowscale : p.gamma;  // always a valid address; rsf folds the loaded value away when there is no scale
    const float rsf = p.rowscale ? 1.0f : 0.0f;

    int next = 0;
    for (; pos < xcnt; pos = next) {
        const int tile = xbase + pos;
        const int m_base = tile * 32;
        if constexpr (LNF) {  // xf holds raw y rows until here (requested one tile ago); first, while few other values are live
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float mu, rs;
                ln_row_frags<NK>(xf[mt], lws, lbs, g, p.eps, mu, rs);
                const int m = min(m_base + mt * 16 + s, p.M - 1);
                if constexpr (SAVE != 0) {
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) st16(p.ln_out + ((int64_t)m * C + ks * 32 + 8 * g) * 2, xf[mt][ks]);
                    if (g == 0) {
                        p.mean[m] = mu;
                        p.rstd[m] = rs;
                    }
                }
            }
        }
        if constexpr (LNF) __builtin_amdgcn_sched_barrier(0);  // keep the loads below out of the block above (C = 96 lives at 256 registers)
        if (dyn) drawn = sched_draw_counted(ctr);  // the tile after this one; read in the last hidden chunk, NCH - 1 chunks of arithmetic from here
        // The tile's residual rows / DropPath scale (consumed in the epilogue) and the next tile's ln fragments are requested in the LAST
        // hidden-chunk iteration: one chunk of MFMA + GELU work and the epilogue cover the round trip, and the 48 + 24 registers
        // they occupy are not live through the whole chunk loop (at C = 96 that spilled, and a scratch reload waits for every
        // outstanding load).  The next fragments go straight into xf, which is dead once the chunk's first product is issued; the
        // last tile fetches itself again (unconditional).
        float4 xrv[MT][CT];
        float rsv[MT];
        f32x4_t o[CT][MT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) o[ct][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto chunk = [&](const int j, auto lastc) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(lastc)::value;
            if constexpr (LAST) {
                __builtin_amdgcn_sched_barrier(0);  // the draw's first use (and the wait the compiler puts in front of it) stays down here
                if (dyn) {  // nothing else of this wave is in flight here: the tile's first stores and the draw are NCH - 1 chunks old
                    next = __builtin_amdgcn_readfirstlane(drawn);
                    reset_ctr = reset_ctr || next == last_draw;
                } else {
                    next = pos + xwaves;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = min(m_base + mt * 16 + s, p.M - 1);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) xrv[mt][ct] = *reinterpret_cast<const float4*>(p.x + (int64_t)m * C + ct * 16 + 4 * g);
                    rsv[mt] = rsp[p.rowscale ? m / p.rps : 0];
                }
            }
            f32x4_t h[4][MT];
            prod_nmajor_b<NK, MT, NK>(h, lds0 + j * PART, s, g, xf);
            if constexpr (LAST) {
                __builtin_amdgcn_sched_barrier(0);  // not above the product that still reads xf
                load_x(xf, next < xcnt ? xbase + next : tile);
            }
            f32x4_t bv[4];
            const uint32_t ba = lds0 + 2 * NCH * PART + (uint32_t)((64 * j + 8 * g) * 4);
            CM_DS_READ128(bv[0], ba, 0);
            CM_DS_READ128(bv[1], ba, 16);
            CM_DS_READ128(bv[2], ba, 128);
            CM_DS_READ128(bv[3], ba, 144);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // element pairs in lockstep (common.hpp "_n" forms): four at a time, two at C = 96 where the two-tile accumulators
            // leave no room for more temporaries (four spilled)
            constexpr int GK = NK == 3 ? 1 : 2;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 4; n2 += GK) {
                    f32x4_t hv[GK];
#pragma unroll
                    for (int k = 0; k < GK; ++k) hv[k] = h[n2 + k][mt];
                    gelu4_bias_n<GK>(hv, &bv[n2]);
#pragma unroll
                    for (int k = 0; k < GK; ++k) h[n2 + k][mt] = hv[k];
                }
            uint4 pf[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                pf[mt][0] = pack8(h[0][mt], h[1][mt]);
                pf[mt][1] = pack8(h[2][mt], h[3][mt]);
            }
            prod_cmajor<NK, MT>(o, lds0 + NCH * PART + j * PART, s, g, pf);
        };
#pragma unroll 1
        for (int j = 0; j < NCH - 1; ++j) chunk(j, BoolC<false>{});
        chunk(NCH - 1, BoolC<true>{});
        // epilogue: b2 / gamma come from LDS, the residual and the scale from the registers requested above
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(m_base + mt * 16 + s, p.M - 1);
            const float rs = fmaf(rsv[mt] - 1.0f, rsf, 1.0f);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + 4 * g;
                const float4 b2 = *reinterpret_cast<const float4*>(b2s + c);
                const float4 gm = *reinterpret_cast<const float4*>(gms + c);
                const float4 xr = xrv[mt][ct];
                const float z0 = o[ct][mt][0] + b2.x, z1 = o[ct][mt][1] + b2.y, z2 = o[ct][mt][2] + b2.z, z3 = o[ct][mt][3] + b2.w;
                if constexpr (SAVE == 1) {
                    uint2 zz;
                    bf16_t* zh = reinterpret_cast<bf16_t*>(&zz);
                    zh[0] = (bf16_t)z0; zh[1] = (bf16_t)z1; zh[2] = (bf16_t)z2; zh[3] = (bf16_t)z3;
                    *reinterpret_cast<uint2*>(p.z + ((int64_t)m * C + c) * 2) = zz;
                }
                *reinterpret_cast<float4*>(p.out + (int64_t)m * C + c) =
                    make_float4(xr.x + rs * gm.x * z0, xr.y + rs * gm.y * z1, xr.z + rs * gm.z * z2, xr.w + rs * gm.w * z3);
            }
        }
    }
    if (reset_ctr && lane == 0) sched_reset(ctr);
}

// DG: the LayerScale gradient dgamma[c] += sum_rows rs g z is formed here from the saved z (p.zin); false (round 4): z is neither saved
// nor read -- dgamma follows from the pwconv2 weight gradient (lnx_layerscale_apply_wgrad)
template <int NK, int MT, bool ST, bool LNB, bool DG>
__global__ __launch_bounds__(512) void convmlp_bwd_res_kernel(const CmP p) {
    constexpr int C = Geo<NK>::C;
    constexpr int CT = Geo<NK>::CT;
    constexpr int PART = Geo<NK>::PART;
    constexpr int NCH = 4 * C / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [W1 image][W2^T image][b1][dgamma partial]
    unsigned char* w1img = smem;
    unsigned char* w2timg = smem + NCH * PART;
    float* b1s = reinterpret_cast<float*>(smem + 2 * NCH * PART);
    float* dgs = b1s + 4 * C;
    float* lws = dgs + C;  // LNB: [C] LayerNorm weight, [2C] column-sum partials
    float* dls = lws + C;
    float* gms = dls + 2 * C;  // [C] LayerScale gamma
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    for (int i = threadIdx.x; i < 4 * C; i += 512) b1s[i] = p.b1[i];
    for (int i = threadIdx.x; i < C; i += 512) {
        dgs[i] = 0.f;
        gms[i] = p.gamma[i];
        if constexpr (LNB) {
            lws[i] = p.lnw[i];
            dls[i] = dls[C + i] = 0.f;
        }
    }
    const bool dyn = p.tile_slot >= 0;  // tile scheduling as in convmlp_fwd_res_kernel
    const TileShare sh = tile_share((p.M + 16 * MT - 1) / (16 * MT));
    unsigned* const ctr = &g_tile_ctr[dyn ? p.tile_slot : 0][sh.part][0];
    int drawn = dyn ? sched_draw_counted(ctr) : 0;
    dma_all_nmajor<NK>(w1img, p.w1, wave, lane);
    dma_all_nmajor<NK>(w2timg, p.w2t, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float dgp[NK][8];  // dgamma partials of this lane's channels ks*32 + 8g + j, summed over its rows
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) dgp[ks][j] = 0.f;

    // Software pipeline over this wave's tiles, no memory operation under a branch (see convmlp_fwd_res_kernel): the operands of tile
    // t + 1 are requested after tile t's hidden-chunk loop, BEFORE its epilogue stores, so their round trip runs under the epilogue's
    // arithmetic and the wait for them does not include those stores' acknowledgements.
    const int ntile = (p.M + 16 * MT - 1) / (16 * MT);
    const float* rsp = p.rowscale ? p.rowscale : p.gamma;
    const float rsf = p.rowscale ? 1.0f : 0.0f;
    struct TileIn {
        uint4 x[MT][NK], z[MT][NK];
        float4 g0[MT][NK], g1[MT][NK];
        float rs[MT];
        LnRow<NK> lr[MT];
    };
    auto fetch_tile = [&](TileIn& t, int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(tile * (16 * MT) + mt * 16 + s, p.M - 1);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int c = ks * 32 + 8 * g;
                t.x[mt][ks] = ld16(p.ln + ((int64_t)m * C + c) * 2);
                t.g0[mt][ks] = *reinterpret_cast<const float4*>(p.g + (int64_t)m * C + c);
                t.g1[mt][ks] = *reinterpret_cast<const float4*>(p.g + (int64_t)m * C + c + 4);
                if constexpr (DG) t.z[mt][ks] = ld16(p.zin + ((int64_t)m * C + c) * 2);
            }
            t.rs[mt] = rsp[p.rowscale ? m / p.rps : 0];
            if constexpr (LNB) ln_bwd_fetch<NK>(t.lr[mt], p, m, g);
        }
    };
    const int xbase = sh.base, xcnt = sh.cnt, xwaves = 8 * sh.workers;
    const int last_draw = xcnt + xwaves - 1;
    int pos = dyn ? __builtin_amdgcn_readfirstlane(drawn) : sh.index * 8 + wave;
    bool reset_ctr = dyn && pos == last_draw;
    TileIn cur;
    if (pos < xcnt) fetch_tile(cur, xbase + pos);
    int next = 0;
    for (; pos < xcnt; pos = next) {
        const int tile = xbase + pos;
        const int m_base = tile * (16 * MT);
        uint4 xf[MT][NK], zf[MT][NK];
        LnRow<NK> lrow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int m = m_base + mt * 16 + s;
            const bool mv = m < p.M;
            if (!mv) m = p.M - 1;  // redoes row M - 1, stores the same bytes as its live twin; masked out of the column sums only
            const float rs = fmaf(cur.rs[mt] - 1.0f, rsf, 1.0f);
            lrow[mt] = cur.lr[mt];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int c = ks * 32 + 8 * g;
                xf[mt][ks] = cur.x[mt][ks];
                const float4 g0 = cur.g0[mt][ks], g1 = cur.g1[mt][ks];
                const float4 a0 = *reinterpret_cast<const float4*>(gms + c), a1 = *reinterpret_cast<const float4*>(gms + c + 4);
                Vec16<bf16_t> zin, dzv, dzs;
                if constexpr (DG) zin.raw = cur.z[mt][ks];
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float sg = rs * gv[j];
                    dzv.set(j, sg * av[j]);
                    dzs.set(j, p.dz_plain ? sg : sg * av[j]);
                    if constexpr (DG) dgp[ks][j] = fmaf(mv ? sg : 0.f, zin.get(j), dgp[ks][j]);
                }
                zf[mt][ks] = dzv.raw;
                st16(p.dz + ((int64_t)m * C + c) * 2, dzs.raw);
            }
        }
        if (dyn) drawn = sched_draw_counted(ctr);  // the tile after this one (behind this tile's dz stores, in front of the chunk loop's)
        f32x4_t dl[CT][MT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dl[ct][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j = 0; j < NCH; ++j) {
            f32x4_t h[4][MT], da[4][MT];
            prod_nmajor_b<NK, MT, NK>(h, lds0 + j * PART, s, g, xf);
            prod_nmajor_b<NK, MT, NK>(da, lds0 + NCH * PART + j * PART, s, g, zf);
            f32x4_t bv[4];
            const uint32_t ba = lds0 + 2 * NCH * PART + (uint32_t)((64 * j + 8 * g) * 4);
            CM_DS_READ128(bv[0], ba, 0);
            CM_DS_READ128(bv[1], ba, 16);
            CM_DS_READ128(bv[2], ba, 128);
            CM_DS_READ128(bv[3], ba, 144);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n2 = 0; n2 < 4; n2 += 2) {  // four element pairs in lockstep
                    f32x4_t hv[2] = {h[n2][mt], h[n2 + 1][mt]}, dv[2] = {da[n2][mt], da[n2 + 1][mt]};
                    gelu4_bias_grad_n<2>(hv, &bv[n2], dv);
                    h[n2][mt] = hv[0];
                    h[n2 + 1][mt] = hv[1];
                    da[n2][mt] = dv[0];
                    da[n2 + 1][mt] = dv[1];
                }
            uint4 pd[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint4 pa0 = pack8(h[0][mt], h[1][mt]), pa1 = pack8(h[2][mt], h[3][mt]);
                pd[mt][0] = pack8(da[0][mt], da[1][mt]);
                pd[mt][1] = pack8(da[2][mt], da[3][mt]);
                const int m = min(m_base + mt * 16 + s, p.M - 1);
                if constexpr (ST) {
                    const int64_t off = ((int64_t)m * (4 * C) + 64 * j + 8 * g) * 2;
                    st16(p.act + off, pa0);
                    st16(p.act + off + 64, pa1);
                    st16(p.dh + off, pd[mt][0]);
                    st16(p.dh + off + 64, pd[mt][1]);
                }
            }
            prod_tr_nmajor<NK, MT>(dl, lds0 + j * PART, s, g, pd);  // dln += dH . W1  (W1 read transposed)
        }
        __builtin_amdgcn_sched_barrier(0);  // the draw's first use stays down here: it is older than every store of the chunk loop and the
                                            // counter is in order, so the wait the compiler puts here lets those stores stay in flight
        if (dyn) {
            next = __builtin_amdgcn_readfirstlane(drawn);
            reset_ctr = reset_ctr || next == last_draw;
        } else {
            next = pos + xwaves;
        }
        fetch_tile(cur, next < xcnt ? xbase + next : tile);  // (the last tile fetches itself again: unconditional)
        if constexpr (LNB) {
            ln_bwd_rows<NK, MT, 0>(p, dl, lrow[0], m_base + s, s, g, lws, dls);
            if constexpr (MT == 2) ln_bwd_rows<NK, MT, 1>(p, dl, lrow[MT - 1], m_base + 16 + s, s, g, lws, dls);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = min(m_base + mt * 16 + s, p.M - 1);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 16 + 4 * g;
                    uint2 v;
                    bf16_t* vh = reinterpret_cast<bf16_t*>(&v);
                    vh[0] = (bf16_t)dl[ct][mt][0]; vh[1] = (bf16_t)dl[ct][mt][1]; vh[2] = (bf16_t)dl[ct][mt][2]; vh[3] = (bf16_t)dl[ct][mt][3];
                    *reinterpret_cast<uint2*>(p.dln + ((int64_t)m * C + c) * 2) = v;
                }
            }
        }
    }
    // dgamma: sum the 16 row-lanes (s) of each g, then LDS atomics across waves, then one global atomic per channel
    if constexpr (DG) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = dgp[ks][j];
                t += __shfl_xor(t, 1, 64);
                t += __shfl_xor(t, 2, 64);
                t += __shfl_xor(t, 4, 64);
                t += __shfl_xor(t, 8, 64);
                if (s == 0) atomicAdd(&dgs[ks * 32 + 8 * g + j], t);
            }
    }
    __syncthreads();
    if constexpr (DG)
        for (int i = threadIdx.x; i < C; i += 512) atomicAdd(p.dgamma + i, dgs[i]);
    if constexpr (LNB)
        for (int i = threadIdx.x; i < 2 * C; i += 512) p.part[(int64_t)blockIdx.x * (2 * C) + i] = dls[i];
    if (reset_ctr && lane == 0) sched_reset(ctr);
}

// dw[c] += sum over workgroups of part[wg][c], db[c] += ... of part[wg][C + c]: one thread per entry and slice of workgroups,
// partial sums in a fixed order, one atomic per slice
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ part, int nwg, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * C) return;
    const int per = (nwg + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int w = w0;
    for (; w + 4 <= w1; w += 4) {
        t0 += part[(int64_t)w * 2 * C + e];
        t1 += part[(int64_t)(w + 1) * 2 * C + e];
        t2 += part[(int64_t)(w + 2) * 2 * C + e];
        t3 += part[(int64_t)(w + 3) * 2 * C + e];
    }
    for (; w < w1; ++w) t0 += part[(int64_t)w * 2 * C + e];
    const float t = (t0 + t1) + (t2 + t3);
    if (w0 < w1) atomicAdd((e < C ? dw : db) + (e < C ? e : e - C), t);
}

// waves per workgroup of the streamed-weight kernels: 8 (two per SIMD share one weight stream) unless LNX_CM_NW=4
static const bool nw8 = !(getenv("LNX_CM_NW") && atoi(getenv("LNX_CM_NW")) == 4);

// workgroups of the backward launch for (C, M) -- ONE place: the launchers below, the LayerNorm partial-sum scratch they check
// (2 C floats per workgroup) and lnx_convmlp_bwd_ws_floats(), which the plan sizes that scratch from
static int bwd_grid(int C, int M) {
    if (C <= 96) {  // resident-weight kernels: 8 waves per workgroup, 16 MT rows per wave tile (MT = 2 at C = 32)
        const int rows = C == 32 ? 32 : 16;
        const int grid = cdiv(cdiv(M, rows), 8);
        return grid > 256 ? 256 : grid;
    }
    return cdiv(M, 16 * (nw8 ? 8 : 4));  // streamed-weight kernels: NW waves x 16 rows per workgroup
}

// workgroups a resident-weight (one per CU) launch may use: the CUs minus the margin, 256 at most (the partial-sum scratch is sized for that)
static int res_room() {
    const int cus = device_cus();
    const int room = persistent_cus(cus > 0 ? cus : 256);
    return room > 256 ? 256 : room;
}

template <int NK, bool LNF, int SAVE>
int launch_fwd_res_t(const CmP& p, hipStream_t st) {
    const size_t lds = 2 * (4 * Geo<NK>::C / 64) * Geo<NK>::PART + 8 * Geo<NK>::C * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_fwd_res_kernel<NK, LNF, SAVE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    int grid = cdiv(cdiv(p.M, 32), 8);
    const int room = res_room();
    if (grid > room) grid = room;
    CmP q = p;
    q.tile_slot = tile_sched_static() ? -1 : tile_slot_of(st);
    hipLaunchKernelGGL((convmlp_fwd_res_kernel<NK, LNF, SAVE>), dim3(grid), dim3(512), lds, st, q);
    return 0;
}
template <int NK>
int launch_fwd_res(const CmP& p, hipStream_t st) {
    if (p.y) return p.z ? launch_fwd_res_t<NK, true, 1>(p, st) : (p.ln_out ? launch_fwd_res_t<NK, true, 2>(p, st) : launch_fwd_res_t<NK, true, 0>(p, st));
    return p.z ? launch_fwd_res_t<NK, false, 1>(p, st) : launch_fwd_res_t<NK, false, 0>(p, st);
}
// LNB launches: the workgroups' column sums (p.part) are folded into the LayerNorm weight / bias gradient right behind the kernel
inline int reduce_ln_partials(const CmP& p, int nwg, hipStream_t st) {
    if (p.ln_defer) {
        ln_postpone_reduce(p.part, nwg, p.C, p.dlnw, p.dlnb, nwg >= 64 ? 64 : 1, st);
        return 0;
    }
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3(cdiv(2 * p.C, 256), nwg >= 64 ? 64 : 1), dim3(256), 0, st, p.part, nwg, p.C, p.dlnw, p.dlnb);
    return 0;
}
template <int NK, int MT, bool ST, bool LNB, bool DG>
int launch_bwd_res_t(const CmP& p, hipStream_t st) {
    const size_t lds = 2 * (4 * Geo<NK>::C / 64) * Geo<NK>::PART + 9 * Geo<NK>::C * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_bwd_res_kernel<NK, MT, ST, LNB, DG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    int grid = bwd_grid(p.C, p.M);
    if (LNB && (int64_t)grid * 2 * p.C > p.part_floats) return -1;  // (checked against the unclamped grid: what lnx_convmlp_bwd_ws_floats reports)
    const int room = res_room();
    if (grid > room) grid = room;
    CmP q = p;
    q.tile_slot = tile_sched_static() ? -1 : tile_slot_of(st);
    hipLaunchKernelGGL((convmlp_bwd_res_kernel<NK, MT, ST, LNB, DG>), dim3(grid), dim3(512), lds, st, q);
    return LNB ? reduce_ln_partials(p, grid, st) : 0;
}
template <int NK, int MT>
int launch_bwd_res(const CmP& p, hipStream_t st) {
    // (the plan's forms: act / dh written; with the block LayerNorm inside and without z, or neither -- the other combinations are
    // for callers of the C entry point and for the A/B switches)
    if (p.zin == nullptr) {
        if (p.y) return p.act ? launch_bwd_res_t<NK, MT, true, true, false>(p, st) : launch_bwd_res_t<NK, MT, false, true, false>(p, st);
        return p.act ? launch_bwd_res_t<NK, MT, true, false, false>(p, st) : launch_bwd_res_t<NK, MT, false, false, false>(p, st);
    }
    if (p.y) return p.act ? launch_bwd_res_t<NK, MT, true, true, true>(p, st) : launch_bwd_res_t<NK, MT, false, true, true>(p, st);
    return p.act ? launch_bwd_res_t<NK, MT, true, false, true>(p, st) : launch_bwd_res_t<NK, MT, false, false, true>(p, st);
}

template <int NK, int MT, int NW, bool LNF>
int launch_fwd_t(const CmP& p, hipStream_t st) {
    const size_t lds = 3 * 2 * Geo<NK>::PART + 4 * Geo<NK>::C * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_fwd_kernel<NK, MT, NW, LNF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((convmlp_fwd_kernel<NK, MT, NW, LNF>), dim3(cdiv(p.M, 16 * NW * MT)), dim3(64 * NW), lds, st, p);
    return 0;
}
template <int NK, int MT, int NW>
int launch_fwd(const CmP& p, hipStream_t st) {
    return p.y ? launch_fwd_t<NK, MT, NW, true>(p, st) : launch_fwd_t<NK, MT, NW, false>(p, st);
}

template <int NK, int MT, int NW, bool ST, bool LNB, bool DG>
int launch_bwd_t(const CmP& p, hipStream_t st) {
    const size_t lds = 2 * 3 * Geo<NK>::PART + 8 * Geo<NK>::C * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_bwd_kernel<NK, MT, NW, ST, LNB, DG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    static_assert(MT == 1, "bwd_grid() assumes 16-row wave tiles in the streamed-weight backward");
    const int grid = bwd_grid(p.C, p.M);
    if (LNB && (int64_t)grid * 2 * p.C > p.part_floats) return -1;
    hipLaunchKernelGGL((convmlp_bwd_kernel<NK, MT, NW, ST, LNB, DG>), dim3(grid), dim3(64 * NW), lds, st, p);
    return LNB ? reduce_ln_partials(p, grid, st) : 0;
}
template <int NK, int MT, int NW>
int launch_bwd(const CmP& p, hipStream_t st) {
    if (p.zin == nullptr) {
        if (p.y) return p.act ? launch_bwd_t<NK, MT, NW, true, true, false>(p, st) : launch_bwd_t<NK, MT, NW, false, true, false>(p, st);
        return p.act ? launch_bwd_t<NK, MT, NW, true, false, false>(p, st) : launch_bwd_t<NK, MT, NW, false, false, false>(p, st);
    }
    if (p.y) return p.act ? launch_bwd_t<NK, MT, NW, true, true, true>(p, st) : launch_bwd_t<NK, MT, NW, false, true, true>(p, st);
    return p.act ? launch_bwd_t<NK, MT, NW, true, false, true>(p, st) : launch_bwd_t<NK, MT, NW, false, false, true>(p, st);
}

}  // namespace

extern "C" int64_t lnx_convmlp_bwd_ws_floats(int C, int M) {
    if (!lnx_convmlp_supported(LNX_BF16, C) || M <= 0) return 0;
    return (int64_t)bwd_grid(C, M) * 2 * C;
}

extern "C" int lnx_convmlp_supported(int dtype, int C) { return dtype == LNX_BF16 && (C == 32 || C == 64 || C == 96 || C == 128 || C == 192); }

extern "C" int lnx_convmlp_fwd(const lnx_convmlp_args* a, void* stream) {
    LNX_CHECK(a && (a->ln || a->y) && a->w1 && a->w2 && a->b1 && a->b2 && a->gamma && a->x && a->out, "lnx_convmlp_fwd: null operand");
    if (a->y) {
        LNX_CHECK(a->ln_w && a->ln_b && a->ln == nullptr, "lnx_convmlp_fwd: the fused LayerNorm form takes y, ln_w, ln_b and no ln");
        LNX_CHECK((a->mean == nullptr) == (a->ln_out == nullptr) && (a->rstd == nullptr) == (a->ln_out == nullptr) && (a->z == nullptr || a->ln_out != nullptr),
                  "lnx_convmlp_fwd: with y, the outputs ln_out, mean and rstd are written together (what a backward needs) or not at all; z only with them");
    }
    LNX_CHECK(lnx_convmlp_supported(a->dtype, a->C), "lnx_convmlp_fwd: unsupported dtype %d / C %d (bf16, C in {32,64,96,128,192})", a->dtype, a->C);
    LNX_CHECK(a->M > 0, "lnx_convmlp_fwd: empty");
    if (a->rowscale) LNX_CHECK(a->rows_per_sample > 0, "lnx_convmlp_fwd: rowscale needs rows_per_sample");
    CmP p{};
    p.ln = (const unsigned char*)a->ln; p.w1 = (const unsigned char*)a->w1; p.w2 = (const unsigned char*)a->w2;
    p.b1 = a->b1; p.b2 = a->b2; p.gamma = a->gamma; p.rowscale = a->rowscale; p.x = a->x; p.out = a->out; p.z = (unsigned char*)a->z;
    p.M = a->M; p.C = a->C; p.rps = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
    p.y = (const unsigned char*)a->y; p.lnw = a->ln_w; p.lnb = a->ln_b; p.eps = a->ln_eps; p.ln_out = (unsigned char*)a->ln_out; p.mean = a->mean; p.rstd = a->rstd;
    hipStream_t st = (hipStream_t)stream;
    switch (a->C) {
        case 32: launch_fwd_res<1>(p, st); break;
        case 64: launch_fwd_res<2>(p, st); break;
        case 96: launch_fwd_res<3>(p, st); break;
        case 128: if (nw8) launch_fwd<4, 1, 8>(p, st); else launch_fwd<4, 1, 4>(p, st); break;
        case 192: if (nw8) launch_fwd<6, 1, 8>(p, st); else launch_fwd<6, 1, 4>(p, st); break;
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_convmlp_bwd(const lnx_convmlp_bwd_args* a, void* stream) {
    LNX_CHECK(a && a->g && a->ln && a->w1 && a->w2t && a->w1t && a->b1 && a->gamma && a->dz && a->dln && (a->dgamma || !a->z),
              "lnx_convmlp_bwd: null operand (z may be NULL: dgamma is then not produced here; with z, dgamma is required)");
    LNX_CHECK((a->act == nullptr) == (a->dh == nullptr), "lnx_convmlp_bwd: act and dh must both be given or both be NULL");
    LNX_CHECK(lnx_convmlp_supported(a->dtype, a->C), "lnx_convmlp_bwd: unsupported dtype %d / C %d", a->dtype, a->C);
    LNX_CHECK(a->M > 0, "lnx_convmlp_bwd: empty");
    if (a->rowscale) LNX_CHECK(a->rows_per_sample > 0, "lnx_convmlp_bwd: rowscale needs rows_per_sample");
    if (a->y) LNX_CHECK(a->ln_w && a->mean && a->rstd && a->d_ln_w && a->d_ln_b && a->ws, "lnx_convmlp_bwd: the fused LayerNorm form needs ln_w, mean, rstd, d_ln_w, d_ln_b and ws");
    CmP p{};
    p.y = (const unsigned char*)a->y; p.lnw = a->ln_w; p.mean = const_cast<float*>(a->mean); p.rstd = const_cast<float*>(a->rstd);
    p.part = a->ws; p.part_floats = a->ws_floats; p.dlnw = a->d_ln_w; p.dlnb = a->d_ln_b; p.ln_defer = a->ln_defer;
    p.g = a->g; p.ln = (const unsigned char*)a->ln; p.zin = (const unsigned char*)a->z; p.w1 = (const unsigned char*)a->w1;
    p.w2t = (const unsigned char*)a->w2t; p.w1t = (const unsigned char*)a->w1t; p.b1 = a->b1; p.gamma = a->gamma; p.rowscale = a->rowscale;
    p.act = (unsigned char*)a->act; p.dh = (unsigned char*)a->dh; p.dz = (unsigned char*)a->dz; p.dln = (unsigned char*)a->dln; p.dgamma = a->dgamma;
    p.M = a->M; p.C = a->C; p.rps = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
    p.dz_plain = a->dz_plain != 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = 0;
    switch (a->C) {
        // rows per wave tile (16 MT): MT = 2 halves the LDS weight reads per MFMA but needs 256+ registers at C >= 64
        // (it spilled 26 / 103 of them to scratch)
        case 32: rc = launch_bwd_res<1, 2>(p, st); break;
        case 64: rc = launch_bwd_res<2, 1>(p, st); break;
        case 96: rc = launch_bwd_res<3, 1>(p, st); break;
        case 128: rc = nw8 ? launch_bwd<4, 1, 8>(p, st) : launch_bwd<4, 1, 4>(p, st); break;
        case 192: rc = nw8 ? launch_bwd<6, 1, 8>(p, st) : launch_bwd<6, 1, 4>(p, st); break;
    }
    LNX_CHECK(rc == 0, "lnx_convmlp_bwd: ws too small for the LayerNorm partial sums (M=%d C=%d ws_floats=%lld)", a->M, a->C, (long long)a->ws_floats);
    LNX_LAUNCH_CHECK();
    return 0;
}
