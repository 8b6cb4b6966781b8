// Global multi-head attention with the reference's cos-only "2D RoPE" for gfx950.
//
// Reference: RoPE2DAttention.forward, standard path (blocks/rope_2d_mhsa.py:422-505):
//   q,k image tokens scaled pairwise by cos(theta) (:176-218, finding F1), q *= d^-0.5
//   (:456), S = q k^T in fp32 (:495), softmax (:496), O = P v (:498), O laid out [B,N,h*d].
// head_dim is 64 in every shipped config.  N is small (52..580), so the whole problem of a
// (batch, head) pair is a few 64-key tiles; parallelism comes from batch x heads x q-tiles.
//
// Layout trick used everywhere below: score tiles are computed TRANSPOSED (keys in the
// accumulator registers, the query -- or in the dK/dV kernel the key -- on the lane), so
// every per-row softmax quantity is lane-local and the probability tile is already the
// B operand of the next MFMA ("accumulator as operand"); the other operand of that second
// product comes from LDS through ds_read_b64_tr_b16 (bf16) or plain b32 reads (fp32).
//
// Kernels:  attn_fwd  -> o, lse
//           attn_bwd_dq   (per 64 queries: delta, dq, cos-gradient part of q)
//           attn_bwd_dkv  (per 64 keys:    dk, dv, cos-gradient part of k)
#include <stdlib.h>

#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int HD = 64;   // head dim
constexpr int BT = 64;   // rows per tile (queries or keys)

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <typename T> struct AT {
    static constexpr int EPV = TT<T>::EPV;
    static constexpr int NKK = sizeof(T);               // 64-byte k-chunks per head row (2 bf16 / 4 fp32)
    static constexpr int ROWB = HD * sizeof(T);         // bytes per row in the row image
    static constexpr int NCH = ROWB / 16;               // 16-byte chunks per row
    static constexpr int TRB = sizeof(T) == 2 ? 160 : 272;  // padded row bytes of the transposed-read image
    static constexpr int ROW_IMG = BT * ROWB;
    static constexpr int TR_IMG = BT * TRB;
};

__device__ __forceinline__ void mfma_bf16(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_f32(f32x4_t& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }

template <typename T> __device__ __forceinline__ void mfma_chunk(f32x4_t& acc, const uint4& a, const uint4& b) {
    if constexpr (sizeof(T) == 2) {
        mfma_bf16(acc, a, b);
    } else {
        const float* fa = reinterpret_cast<const float*>(&a);
        const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_f32(acc, fa[j], fb[j]);
    }
}

__device__ __forceinline__ uint2 tr_read(const unsigned char* p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    return __builtin_bit_cast(uint2, v);
}

// One 16-byte chunk (EPV consecutive head channels starting at d0) of token n for the given operand column base,
// optionally multiplied pairwise by cos and by `scale`.  Split in two so that a caller can issue every fetch of a
// batch before the first use: fetch() is unconditional (a row beyond N reads row N - 1, a class token reads the first
// cos row) -- a load under `if (n < N)` whose result is merged with a zero is waited for on the spot, which turns a
// staging loop into one memory round trip per chunk.
template <typename T, bool COS> struct Chunk {
    uint4 raw;
    float cf[COS ? TT<T>::EPV / 2 : 1];
    __device__ __forceinline__ void fetch(const T* __restrict__ base, int64_t ld, int n, int N, int E, int d0, const float* __restrict__ cos_tab, int heads, int head) {
        const int nc = min(n, N - 1);
        raw = ld16(base + (int64_t)nc * ld + d0);
        if constexpr (COS) {
            const float* cp = cos_tab + ((int64_t)max(nc - E, 0) * heads + head) * 32 + (d0 >> 1);
            if constexpr (TT<T>::EPV == 8) {
                const float4 c4 = *reinterpret_cast<const float4*>(cp);
                cf[0] = c4.x; cf[1] = c4.y; cf[2] = c4.z; cf[3] = c4.w;
            } else {
                const float2 c2 = *reinterpret_cast<const float2*>(cp);
                cf[0] = c2.x; cf[1] = c2.y;
            }
        }
    }
    __device__ __forceinline__ uint4 value(int n, int N, int E, float scale) const {
        if (n >= N) return make_uint4(0, 0, 0, 0);
        Vec16<T> v;
        v.raw = raw;
        if constexpr (COS) {
            constexpr int EPV = TT<T>::EPV;
            if (n >= E) {
#pragma unroll
                for (int j = 0; j < EPV; ++j) v.set(j, v.get(j) * cf[j >> 1] * scale);
            } else if (scale != 1.0f) {
#pragma unroll
                for (int j = 0; j < EPV; ++j) v.set(j, v.get(j) * scale);
            }
        }
        return v.raw;
    }
};
template <typename T, bool COS>
__device__ __forceinline__ uint4 load_chunk(const T* __restrict__ base, int64_t ld, int n, int N, int E, int d0, const float* __restrict__ cos_tab,
                                            int heads, int head, float scale) {
    Chunk<T, COS> c;
    c.fetch(base, ld, n, N, E, d0, cos_tab, heads, head);
    return c.value(n, N, E, scale);
}

// Stage a 64-row tile (tokens n0..n0+63) into LDS: swizzled row image (16-byte fragment
// reads, rows = MFMA rows) and/or padded image for transposed reads.
template <typename T, bool COS, bool ROWIMG, bool TRIMG, int NT = 256>
__device__ __forceinline__ void stage_tile(unsigned char* rowimg, unsigned char* trimg, const T* __restrict__ base, int64_t ld, int n0, int N, int E,
                                           const float* __restrict__ cos_tab, int heads, int head, float scale) {
    constexpr int NCH = AT<T>::NCH;
    constexpr int EPV = AT<T>::EPV;
    for (int i = threadIdx.x; i < BT * NCH; i += NT) {
        const int r = i / NCH, c = i % NCH;
        const uint4 v = load_chunk<T, COS>(base, ld, n0 + r, N, E, c * EPV, cos_tab, heads, head, scale);
        if constexpr (ROWIMG) st16(rowimg + r * AT<T>::ROWB + ((c ^ (r & 7)) << 4), v);
        if constexpr (TRIMG) st16(trimg + r * AT<T>::TRB + c * 16, v);
    }
}

// The same staging split in two, for the tiled kernels' software pipeline: fetch() requests a 64-row tile into registers
// (unconditional, clamped rows -- call it for min(next, last) rather than under `if (more)`), commit() writes it to the LDS
// images one loop trip later, after the products of the current tile have been issued in between.
template <typename T, bool COS, int NT>
struct TileFetch {
    static constexpr int NCH = AT<T>::NCH, EPV = AT<T>::EPV, PER = BT * NCH / NT;
    Chunk<T, COS> ch[PER];
    __device__ __forceinline__ void fetch(const T* __restrict__ base, int64_t ld, int n0, int N, int E, const float* __restrict__ cos_tab, int heads, int head) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = threadIdx.x + u * NT;
            ch[u].fetch(base, ld, n0 + i / NCH, N, E, (i % NCH) * EPV, cos_tab, heads, head);
        }
    }
    template <bool ROWIMG, bool TRIMG>
    __device__ __forceinline__ void commit(unsigned char* rowimg, unsigned char* trimg, int n0, int N, int E, float scale) const {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = threadIdx.x + u * NT;
            const int r = i / NCH, c = i % NCH;
            const uint4 v = ch[u].value(n0 + r, N, E, scale);
            if constexpr (ROWIMG) st16(rowimg + r * AT<T>::ROWB + ((c ^ (r & 7)) << 4), v);
            if constexpr (TRIMG) st16(trimg + r * AT<T>::TRB + c * 16, v);
        }
    }
};

// fragment of a row operand held in registers: lane (s, g) <- token row, chunks kk*4 + g
template <typename T, bool COS>
__device__ __forceinline__ void load_row_frag(uint4 (&f)[AT<T>::NKK], const T* __restrict__ base, int64_t ld, int n, int N, int E, int g,
                                              const float* __restrict__ cos_tab, int heads, int head, float scale) {
    Chunk<T, COS> c[AT<T>::NKK];
#pragma unroll
    for (int kk = 0; kk < AT<T>::NKK; ++kk) c[kk].fetch(base, ld, n, N, E, (kk * 4 + g) * AT<T>::EPV, cos_tab, heads, head);
#pragma unroll
    for (int kk = 0; kk < AT<T>::NKK; ++kk) f[kk] = c[kk].value(n, N, E, scale);
}

// acc[t] (t = 0..3, 16 rows each) = Rows(img, row0 + 16 t + s) . frag^T over the 64 channels
// nt = number of 16-row groups of the tile that hold real tokens (the rest is padding: skipped, acc = 0)
template <typename T>
__device__ __forceinline__ void rows_times_frag(f32x4_t (&acc)[4], const unsigned char* rowimg, int s, int g, const uint4 (&frag)[AT<T>::NKK], int nt = 4) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (t >= nt) continue;
        const int row = t * 16 + s;
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            const uint4 a = ld16(rowimg + row * AT<T>::ROWB + (((kk * 4 + g) ^ (row & 7)) << 4));
            mfma_chunk<T>(acc[t], a, frag[kk]);
        }
    }
}

// out[dt] += Img^T . P   where P[t][r] holds, for the lane's column, the value of tile row
// 16 t + 4 g + r (t = 0..3) -- i.e. contraction over the 64 tile rows.
// Same product from the PADDED image (rows of TRB = 160 bytes, unswizzled): at that pitch the 16-byte fragment reads
// of every 16-lane group of a ds_read_b128 also fall on 16 distinct bank slots, so one image serves both the row
// fragments and the transposed reads (half the LDS of keeping a swizzled row image beside it).
template <typename T>
__device__ __forceinline__ void rows_times_frag_pad(f32x4_t (&acc)[4], const unsigned char* img, int s, int g, const uint4 (&frag)[AT<T>::NKK], int nt) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (t >= nt) continue;
        const int row = t * 16 + s;
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            const uint4 a = ld16(img + row * AT<T>::TRB + ((kk * 4 + g) << 4));
            mfma_chunk<T>(acc[t], a, frag[kk]);
        }
    }
}

// nt as above: 32-row contraction steps made of padding only are skipped
template <typename T>
__device__ __forceinline__ void imgT_times_regs(f32x4_t (&out)[4], const unsigned char* trimg, int s, int g, const float (&p)[4][4], int nt = 4) {
    if constexpr (sizeof(T) == 2) {
        uint4 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Vec16<bf16_t> v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, p[2 * ks + (j >> 2)][j & 3]);
            pf[ks] = v.raw;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (2 * ks >= nt) continue;
            const int r0 = ks * 32 + 4 * g + (s >> 2);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int col = (dt * 16 + 4 * (s & 3)) * 2;
                const uint2 a0 = tr_read(trimg + r0 * AT<T>::TRB + col);
                const uint2 a1 = tr_read(trimg + (r0 + 16) * AT<T>::TRB + col);
                mfma_bf16(out[dt], make_uint4(a0.x, a0.y, a1.x, a1.y), pf[ks]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= nt) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + 4 * g + r;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float a = *reinterpret_cast<const float*>(trimg + row * AT<T>::TRB + (dt * 16 + s) * 4);
                    mfma_f32(out[dt], a, p[t][r]);
                }
            }
        }
    }
}

// ---- the same two products with the number of live 16-row groups as a compile-time constant -------------------------------
// With a run-time `nt` every group sits under its own branch: hipcc then emits `ds_read; s_waitcnt lgkmcnt(0); v_mfma` per
// group -- one exposed LDS round trip per pair of MFMAs (round 3: the resident backward kernels spent most of their key /
// query loop that way).  The images are staged in units of 32 rows.  The forward uses NT = 4 for a tile with more than 32 live rows
// (a fourth group of pure padding costs two MFMAs) and NT = 2 for the last tile otherwise; the backward kernels walk in steps of
// NT = 2 throughout (half the live accumulators per step: no scratch).
// PHASE_FENCE keeps the products of one tile step apart: left free, the scheduler hoists the fragment loads of all of them to
// the top and the 128-register budget (two workgroups per CU) spills.
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int V> struct IC { static constexpr int value = V; };
template <typename T, int NT>
__device__ __forceinline__ void rows_times_frag_pad_n(f32x4_t (&acc)[4], const unsigned char* img, int s, int g, const uint4 (&frag)[AT<T>::NKK]) {
    uint4 a[NT][AT<T>::NKK];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) a[t][kk] = ld16(img + (t * 16 + s) * AT<T>::TRB + ((kk * 4 + g) << 4));
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) mfma_chunk<T>(acc[t], a[t][kk], frag[kk]);
}
template <typename T, int NT>
__device__ __forceinline__ void rows_times_frag_n(f32x4_t (&acc)[4], const unsigned char* rowimg, int s, int g, const uint4 (&frag)[AT<T>::NKK]) {
    uint4 a[NT][AT<T>::NKK];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) a[t][kk] = ld16(rowimg + (t * 16 + s) * AT<T>::ROWB + (((kk * 4 + g) ^ ((t * 16 + s) & 7)) << 4));
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) mfma_chunk<T>(acc[t], a[t][kk], frag[kk]);
}
// p[t] of the groups t >= NT must be zero (the image rows behind them may lie beyond the staged part only for t >= 2 ceil(NT / 2))
template <typename T, int NT>
__device__ __forceinline__ void imgT_times_regs_n(f32x4_t (&out)[4], const unsigned char* trimg, int s, int g, const float (&p)[4][4]) {
    if constexpr (sizeof(T) == 2) {
        constexpr int NKS = (NT + 1) / 2;
        uint4 pf[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            Vec16<bf16_t> v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, p[2 * ks + (j >> 2)][j & 3]);
            pf[ks] = v.raw;
        }
        uint2 a0[NKS][4], a1[NKS][4];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int r0 = ks * 32 + 4 * g + (s >> 2);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int col = (dt * 16 + 4 * (s & 3)) * 2;
                a0[ks][dt] = tr_read(trimg + r0 * AT<T>::TRB + col);
                a1[ks][dt] = tr_read(trimg + (r0 + 16) * AT<T>::TRB + col);
            }
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mfma_bf16(out[dt], make_uint4(a0[ks][dt].x, a0[ks][dt].y, a1[ks][dt].x, a1[ks][dt].y), pf[ks]);
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = t * 16 + 4 * g + r;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float a = *reinterpret_cast<const float*>(trimg + row * AT<T>::TRB + (dt * 16 + s) * 4);
                    mfma_f32(out[dt], a, p[t][r]);
                }
            }
    }
}

// four consecutive elements in one store (8 bytes of bf16 / 16 bytes of fp32); p is 8-/16-byte aligned
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) {
    if constexpr (sizeof(T) == 2) {
        uint2 r;
        T* h = reinterpret_cast<T*>(&r);
        h[0] = from_f<T>(a);
        h[1] = from_f<T>(b);
        h[2] = from_f<T>(c);
        h[3] = from_f<T>(d);
        *reinterpret_cast<uint2*>(p) = r;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    }
}

// One 64-element head row held as v[dt][0..3] = elements 16 dt + 4 g .. + 3 by the four lanes g of a row (the layout every
// kernel here ends with).  fp32: four 16-byte stores.  bf16: lanes g and g ^ 1 swap halves first, so that the even one writes
// elements 16 dt + 4 g .. + 7 of dt = 0, 2 and the odd one those of dt = 1, 3 -- two 16-byte stores a lane instead of four 8-byte
// ones (whose half-filled 32-byte sectors made the key-side backward write 1.4x its output).  EVERY lane of the wave must call
// (the exchange is a cross-lane operation); `live` gates the stores only.
template <typename T> __device__ __forceinline__ void store_row64(T* rowp, const float (&v)[4][4], int g, bool live) {
    if constexpr (sizeof(T) == 4) {
        if (live) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<float4*>(rowp + dt * 16 + 4 * g) = make_float4(v[dt][0], v[dt][1], v[dt][2], v[dt][3]);
        }
    } else {
        uint2 pk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            T* h = reinterpret_cast<T*>(&pk[dt]);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = from_f<T>(v[dt][j]);
        }
        const bool even = (g & 1) == 0;
        const uint2 s0 = even ? pk[1] : pk[0], s1 = even ? pk[3] : pk[2];
        uint2 r0, r1;
        r0.x = (uint32_t)__shfl_xor((int)s0.x, 16, 64); r0.y = (uint32_t)__shfl_xor((int)s0.y, 16, 64);
        r1.x = (uint32_t)__shfl_xor((int)s1.x, 16, 64); r1.y = (uint32_t)__shfl_xor((int)s1.y, 16, 64);
        if (live) {
            if (even) {
                st16(rowp + 4 * g, make_uint4(pk[0].x, pk[0].y, r0.x, r0.y));
                st16(rowp + 32 + 4 * g, make_uint4(pk[2].x, pk[2].y, r1.x, r1.y));
            } else {
                st16(rowp + 16 + 4 * (g - 1), make_uint4(r0.x, r0.y, pk[1].x, pk[1].y));
                st16(rowp + 48 + 4 * (g - 1), make_uint4(r1.x, r1.y, pk[3].x, pk[3].y));
            }
        }
    }
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lanes s, s+16, s+32, s+48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// sum over the 16 lanes of a DPP row (lanes s = 0..15 of one g): every lane ends with the total
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    return v;
}

// Gradient of the learnable RoPE frequencies, reduced where it is produced (round 3; it used to travel through a
// [2][B, N-E, heads, 32] fp32 tensor, ~66 MB per block, written 4 bytes at a time and read back by a separate kernel).
// Autograd of compute_mixed_cis / apply_rotary_emb through the real part only (finding F1):
//   dfreqs[a, h, j] = sum_{b, n} (d cos(theta[n,h,j]) / d freqs[a,h,j]) * gpair[b, n, h, j],   gpair = dQ~[2j] q[2j] + dQ~[2j+1] q[2j+1] (+ the k term)
// This lane holds the 8 pair gradients gp[dt][pr] (j = 8 dt + 2 g + pr) of ONE row and the matching table entries
// dx / dy = -t_x sin(theta), -t_y sin(theta); rows are summed over the 16 lanes of the DPP row, then added to the
// workgroup's 64 LDS accumulators [a][j] (order of the LDS float adds is not fixed: last-bit differences from run to run).
__device__ __forceinline__ void freq_accum(float* fl, const float (&gp)[4][2], const float (&dx)[4][2], const float (&dy)[4][2], int s, int g) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const float tx = row16_sum(gp[dt][pr] * dx[dt][pr]);
            const float ty = row16_sum(gp[dt][pr] * dy[dt][pr]);
            if (s == 0) {
                const int j = 8 * dt + 2 * g + pr;
                atomicAdd(fl + j, tx);
                atomicAdd(fl + 32 + j, ty);
            }
        }
}
// after the workgroup's last freq_accum: publish (dq kernel) or add to (dk/dv kernel, same grid, launched after it) the partial
template <bool ADD> __device__ __forceinline__ void freq_flush(const float* fl, float* fpart) {
    __syncthreads();
    if (threadIdx.x < 64) {
        float* dst = fpart + (int64_t)blockIdx.x * 64 + threadIdx.x;
        *dst = ADD ? *dst + fl[threadIdx.x] : fl[threadIdx.x];
    }
}

// exp of the softmax: the bf16 kernels use the hardware exponential (v_exp_f32, ~1 ulp), the fp32 (strict-parity) ones libm's
template <typename T> __device__ __forceinline__ float fexp(float x) {
    if constexpr (sizeof(T) == 2) return __expf(x);
    else return expf(x);
}
// exp(x - m) with m fixed along a row.  bf16 kernels: log2(e) folded into the subtraction, one FMA in front of v_exp_f32
// (prep(m) = m log2(e) once per row) instead of subtract + multiply; fp32 (strict parity) kernels: libm.  `clamped` bounds the
// argument by 0 -- a no-op for real (query, key) pairs of the backward (x <= lse), it keeps padding entries finite.
template <typename T> struct RowExp {
    static constexpr float L2E = 1.44269504088896340736f;
    static __device__ __forceinline__ float prep(float m) {
        if constexpr (sizeof(T) == 2) return m * L2E;
        else return m;
    }
    static __device__ __forceinline__ float sub(float x, float mp) {
        if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(fmaf(x, L2E, -mp));
        else return expf(x - mp);
    }
    static __device__ __forceinline__ float clamped(float x, float mp) {
        if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(fminf(fmaf(x, L2E, -mp), 0.f));
        else return expf(fminf(x - mp, 0.f));
    }
};

struct AttnP {
    const void* qkv;
    const float* cos_tab;
    void* o;
    float* lse;
    const void* d_o;
    void* dqkv;
    float* fpart;        // [workgroups][2][32]: per-workgroup partial sums of the freqs gradient (dq kernel writes, dk/dv kernel adds)
    const float* dsin;   // [2][N-E, heads, 32]: -t_x sin(theta), -t_y sin(theta) = d cos(theta) / d freqs[a]
    float* delta;
    int B, N, E, heads;
    int qtiles;
    const unsigned char* amask = nullptr;  // DROP kernels: keep mask [B, heads, N, Np] of the attention probabilities
    float a_inv_keep = 1.0f;
    int Np = 0;                            // N rounded up to a multiple of 64
};

// dk / dv epilogue of one 16-key wave tile: every load in one batch (clamped rows, unconditional), 8-/16-byte stores, the k part
// of the freqs gradient.  Shared by the resident and the tiled kernel.
template <typename T>
__device__ __forceinline__ void dkv_epilogue(const AttnP& p, const f32x4_t (&dk)[4], const f32x4_t (&dv)[4], const T* __restrict__ kb, int64_t ld, int C,
                                             int b, int head, int key, int s, int g, float* fl) {
    const int kc = min(key, p.N - 1);
    const T* kraw = kb + (int64_t)kc * ld;
    const float* cpr = p.cos_tab + ((int64_t)max(kc - p.E, 0) * p.heads + head) * 32;
    const float* sxp = p.dsin + ((int64_t)max(kc - p.E, 0) * p.heads + head) * 32;
    const float* syp = sxp + (int64_t)(p.N - p.E) * p.heads * 32;
    float kr[4][4], cr[4][2], gp[4][2], dx[4][2], dy[4][2];
    const bool rope = p.E < p.N;  // uniform: without image tokens there are no tables
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int d0 = dt * 16 + 4 * g;
        if constexpr (sizeof(T) == 2) {
            const uint2 r = *reinterpret_cast<const uint2*>(kraw + d0);
            const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
#pragma unroll
            for (int j = 0; j < 4; ++j) kr[dt][j] = (float)h[j];
        } else {
            const float4 r = *reinterpret_cast<const float4*>(kraw + d0);
            kr[dt][0] = r.x; kr[dt][1] = r.y; kr[dt][2] = r.z; kr[dt][3] = r.w;
        }
        cr[dt][0] = cr[dt][1] = 1.0f;
        dx[dt][0] = dx[dt][1] = dy[dt][0] = dy[dt][1] = 0.f;
        if (rope) {
            const float2 c2 = *reinterpret_cast<const float2*>(cpr + (d0 >> 1));
            cr[dt][0] = c2.x; cr[dt][1] = c2.y;
            const float2 a2 = *reinterpret_cast<const float2*>(sxp + (d0 >> 1));
            const float2 b2 = *reinterpret_cast<const float2*>(syp + (d0 >> 1));
            dx[dt][0] = a2.x; dx[dt][1] = a2.y;
            dy[dt][0] = b2.x; dy[dt][1] = b2.y;
        }
    }
    const bool row = key < p.N, img = row && key >= p.E;
    T* dkp = reinterpret_cast<T*>(p.dqkv) + ((int64_t)b * p.N + kc) * ld + C + head * HD;
    {
        float ok[4][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const float c0 = img ? cr[dt][0] : 1.0f, c1 = img ? cr[dt][1] : 1.0f;
            ok[dt][0] = dk[dt][0] * c0; ok[dt][1] = dk[dt][1] * c0; ok[dt][2] = dk[dt][2] * c1; ok[dt][3] = dk[dt][3] * c1;
            gp[dt][0] = img ? dk[dt][0] * kr[dt][0] + dk[dt][1] * kr[dt][1] : 0.f;
            gp[dt][1] = img ? dk[dt][2] * kr[dt][2] + dk[dt][3] * kr[dt][3] : 0.f;
        }
        store_row64<T>(dkp, ok, g, row);
    }
    {
        float ov[4][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[dt][j] = dv[dt][j];
        store_row64<T>(dkp + C, ov, g, row);
    }
    if (rope) freq_accum(fl, gp, dx, dy, s, g);
}
// dq epilogue of one 16-query wave tile (the tiled kernel; the resident one fetches its operands ahead of the key loop)
template <typename T>
__device__ __forceinline__ void dq_epilogue(const AttnP& p, const f32x4_t (&dq)[4], const T* __restrict__ qb, int64_t ld, int b, int head, int q, int s, int g,
                                            float scale, float* fl) {
    const int qc = min(q, p.N - 1);
    const T* qraw = qb + (int64_t)qc * ld;
    const float* cpr = p.cos_tab + ((int64_t)max(qc - p.E, 0) * p.heads + head) * 32;
    const float* sxp = p.dsin + ((int64_t)max(qc - p.E, 0) * p.heads + head) * 32;
    const float* syp = sxp + (int64_t)(p.N - p.E) * p.heads * 32;
    float qr[4][4], cr[4][2], gp[4][2], dx[4][2], dy[4][2];
    const bool rope = p.E < p.N;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int d0 = dt * 16 + 4 * g;
        if constexpr (sizeof(T) == 2) {
            const uint2 r = *reinterpret_cast<const uint2*>(qraw + d0);
            const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
#pragma unroll
            for (int j = 0; j < 4; ++j) qr[dt][j] = (float)h[j];
        } else {
            const float4 r = *reinterpret_cast<const float4*>(qraw + d0);
            qr[dt][0] = r.x; qr[dt][1] = r.y; qr[dt][2] = r.z; qr[dt][3] = r.w;
        }
        cr[dt][0] = cr[dt][1] = 1.0f;
        dx[dt][0] = dx[dt][1] = dy[dt][0] = dy[dt][1] = 0.f;
        if (rope) {
            const float2 c2 = *reinterpret_cast<const float2*>(cpr + (d0 >> 1));
            cr[dt][0] = c2.x; cr[dt][1] = c2.y;
            const float2 a2 = *reinterpret_cast<const float2*>(sxp + (d0 >> 1));
            const float2 b2 = *reinterpret_cast<const float2*>(syp + (d0 >> 1));
            dx[dt][0] = a2.x; dx[dt][1] = a2.y;
            dy[dt][0] = b2.x; dy[dt][1] = b2.y;
        }
    }
    const bool row = q < p.N, img = row && q >= p.E;
    T* dqp = reinterpret_cast<T*>(p.dqkv) + ((int64_t)b * p.N + qc) * ld + head * HD;
    float oq[4][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const float c0 = img ? cr[dt][0] * scale : scale, c1 = img ? cr[dt][1] * scale : scale;
        oq[dt][0] = dq[dt][0] * c0; oq[dt][1] = dq[dt][1] * c0; oq[dt][2] = dq[dt][2] * c1; oq[dt][3] = dq[dt][3] * c1;
        gp[dt][0] = img ? scale * (dq[dt][0] * qr[dt][0] + dq[dt][1] * qr[dt][1]) : 0.f;
        gp[dt][1] = img ? scale * (dq[dt][2] * qr[dt][2] + dq[dt][3] * qr[dt][3]) : 0.f;
    }
    store_row64<T>(dqp, oq, g, row);
    if (rope) freq_accum(fl, gp, dx, dy, s, g);
}

// ---------------------------------------------------------------------------------
// forward: one workgroup = 16 NW queries of one (b, head); wave = 16 queries (lane s).  NW = 8 (bf16, long sequences): every
// staged 64-key tile serves 128 queries, i.e. half the staging work, LDS writes and barriers per query of NW = 4.
// p.qtiles = ceil(N / (16 NW)).
// ---------------------------------------------------------------------------------
// DROP: dropout on the attention probabilities (rope_2d_mhsa.py:497), applied after the normalisation: the running sum
// takes the undropped exponentials, P . V the dropped ones
template <typename T, int NW = 4, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kimg = smem;                       // K~ row image
    unsigned char* vimg = smem + AT<T>::ROW_IMG;      // V transposed-read image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int qt = blockIdx.x % p.qtiles;
    const int bh = blockIdx.x / p.qtiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const float scale = 0.125f;  // 64^-0.5

    const int q = qt * (16 * NW) + wave * 16 + s;
    uint4 qf[AT<T>::NKK];
    load_row_frag<T, true>(qf, qb, ld, q, p.N, p.E, g, p.cos_tab, p.heads, head, scale);

    f32x4_t oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const unsigned char* mrow = nullptr;
    if constexpr (DROP) mrow = p.amask + (((int64_t)b * p.heads + head) * p.N + min(q, p.N - 1)) * p.Np + 4 * g;

    const int nkt = (p.N + BT - 1) / BT;
    TileFetch<T, true, 64 * NW> fk;   // the next key tile travels in registers while this one is multiplied
    TileFetch<T, false, 64 * NW> fv;
    fk.fetch(kb, ld, 0, p.N, p.E, p.cos_tab, p.heads, head);
    fv.fetch(vb, ld, 0, p.N, p.E, nullptr, p.heads, head);
    for (int kt = 0; kt < nkt; ++kt) {
        uint32_t mk[4] = {0, 0, 0, 0};
        if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mk[t] = *reinterpret_cast<const uint32_t*>(mrow + kt * BT + 16 * t);
        }
        __syncthreads();
        fk.template commit<true, false>(kimg, nullptr, kt * BT, p.N, p.E, 1.0f);
        fv.template commit<false, true>(nullptr, vimg, kt * BT, p.N, p.E, 1.0f);
        const int nxt = min(kt + 1, nkt - 1) * BT;
        fk.fetch(kb, ld, nxt, p.N, p.E, p.cos_tab, p.heads, head);
        fv.fetch(vb, ld, nxt, p.N, p.E, nullptr, p.heads, head);
        __syncthreads();
        f32x4_t sacc[4];
        rows_times_frag_n<T, 4>(sacc, kimg, s, g, qf);  // S^T[key = 16t + 4g + r][query = s]
        float pv[4][4];
        float mx = -INFINITY;
        if (kt * BT + BT <= p.N) {  // uniform: only the last tile of a sequence can hold padding keys
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[t][r] = sacc[t][r];
                    mx = fmaxf(mx, sacc[t][r]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * BT + t * 16 + 4 * g + r;
                    const float v = key < p.N ? sacc[t][r] : -INFINITY;
                    pv[t][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = group_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float mp = RowExp<T>::prep(m_new);
        const float alpha = RowExp<T>::sub(m_run, mp);
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = RowExp<T>::sub(pv[t][r], mp);
                psum += e;
                pv[t][r] = DROP ? (((mk[t] >> (8 * r)) & 0xffu) ? e * p.a_inv_keep : 0.f) : e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
        imgT_times_regs_n<T, 4>(oacc, vimg, s, g, pv);  // O^T[d = 16dt + 4g + r][query = s]
    }
    const float l_tot = group_sum(l_run);
    const float inv = 1.0f / l_tot;
    {
        float ov[4][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[dt][r] = oacc[dt][r] * inv;
        store_row64<T>(reinterpret_cast<T*>(p.o) + ((int64_t)b * p.N + min(q, p.N - 1)) * C + head * HD, ov, g, q < p.N);
        if (q < p.N && g == 0 && p.lse) p.lse[((int64_t)b * p.heads + head) * p.N + q] = m_run + logf(l_tot);
    }
}

// ---------------------------------------------------------------------------------
// backward, query side: delta, dq (and the q part of the cos gradient)
// ---------------------------------------------------------------------------------
template <typename T, int NW = 4, bool DROP = false>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* kimg = smem;                                        // K~ rows
    unsigned char* vimg = smem + AT<T>::ROW_IMG;                       // V rows
    unsigned char* ktr = smem + 2 * AT<T>::ROW_IMG;                    // K~ transposed-read image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int qt = blockIdx.x % p.qtiles;
    const int bh = blockIdx.x / p.qtiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.N * C + head * HD;
    const T* ob = reinterpret_cast<const T*>(p.o) + (int64_t)b * p.N * C + head * HD;
    const float scale = 0.125f;

    __shared__ float fl[64];  // this workgroup's freqs-gradient partial [a][j]; zeroed here, published behind the loop's barriers
    if (threadIdx.x < 64) fl[threadIdx.x] = 0.f;
    const int q = qt * (16 * NW) + wave * 16 + s;
    uint4 qf[AT<T>::NKK], dof[AT<T>::NKK];
    load_row_frag<T, true>(qf, qb, ld, q, p.N, p.E, g, p.cos_tab, p.heads, head, scale);
    load_row_frag<T, false>(dof, dob, C, q, p.N, p.E, g, nullptr, p.heads, head, 1.0f);
    // delta = sum_d dO * O  (each lane holds 1/4 of the row)
    float dl = 0.f;
    {
        uint4 of[AT<T>::NKK];
        load_row_frag<T, false>(of, ob, C, q, p.N, p.E, g, nullptr, p.heads, head, 1.0f);
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            Vec16<T> a, bb;
            a.raw = dof[kk];
            bb.raw = of[kk];
#pragma unroll
            for (int j = 0; j < AT<T>::EPV; ++j) dl += a.get(j) * bb.get(j);
        }
    }
    const float delta = group_sum(dl);
    const float lse_raw = p.lse[((int64_t)b * p.heads + head) * p.N + min(q, p.N - 1)];  // unconditional load, select after
    const float lse = RowExp<T>::prep(q < p.N ? lse_raw : 0.f);
    if (q < p.N && g == 0) p.delta[((int64_t)b * p.heads + head) * p.N + q] = delta;

    f32x4_t dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned char* mrow = nullptr;
    if constexpr (DROP) mrow = p.amask + (((int64_t)b * p.heads + head) * p.N + min(q, p.N - 1)) * p.Np + 4 * g;

    const int nkt = (p.N + BT - 1) / BT;
    TileFetch<T, true, 64 * NW> fk;   // the next key tile travels in registers while this one is multiplied
    TileFetch<T, false, 64 * NW> fv;
    fk.fetch(kb, ld, 0, p.N, p.E, p.cos_tab, p.heads, head);
    fv.fetch(vb, ld, 0, p.N, p.E, nullptr, p.heads, head);
    for (int kt = 0; kt < nkt; ++kt) {
        uint32_t mk[4] = {0, 0, 0, 0};
        if constexpr (DROP) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mk[t] = *reinterpret_cast<const uint32_t*>(mrow + kt * BT + 16 * t);
        }
        __syncthreads();
        fk.template commit<true, true>(kimg, ktr, kt * BT, p.N, p.E, 1.0f);
        fv.template commit<true, false>(vimg, nullptr, kt * BT, p.N, p.E, 1.0f);
        const int nxt = min(kt + 1, nkt - 1) * BT;
        fk.fetch(kb, ld, nxt, p.N, p.E, p.cos_tab, p.heads, head);
        fv.fetch(vb, ld, nxt, p.N, p.E, nullptr, p.heads, head);
        __syncthreads();
        // no masks: padding keys have zero rows in all three images (their finite dS meets a zero row of K~), padding queries
        // are not stored; the clamp keeps exp finite where (query, key) is not a real pair
        // two half steps of 32 keys: the same products with half the live registers (see the resident kernels)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4_t sacc[4], dpacc[4];
            rows_times_frag_n<T, 2>(sacc, kimg + h * 32 * AT<T>::ROWB, s, g, qf);    // S^T[key][q]
            PHASE_FENCE();
            rows_times_frag_n<T, 2>(dpacc, vimg + h * 32 * AT<T>::ROWB, s, g, dof);  // dP^T[key][q] = V[key] . dO[q]
            PHASE_FENCE();
            float ds[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (t < 2) {
                        const float pr = RowExp<T>::clamped(sacc[t][r], lse);
                        const float dp = DROP ? (((mk[2 * h + t] >> (8 * r)) & 0xffu) ? dpacc[t][r] * p.a_inv_keep : 0.f) : dpacc[t][r];
                        ds[t][r] = pr * (dp - delta);
                    } else {
                        ds[t][r] = 0.f;
                    }
                }
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dq, ktr + h * 32 * AT<T>::TRB, s, g, ds);  // dQ~^T[d][q] += K~^T . dS^T
            PHASE_FENCE();
        }
    }
    dq_epilogue<T>(p, dq, qb, ld, b, head, q, s, g, scale, fl);
    if (p.E < p.N) freq_flush<false>(fl, p.fpart);
}

// ---------------------------------------------------------------------------------
// backward, key side: dk, dv (and the k part of the cos gradient)
// wave = 16 keys (lane s); loops over 64-query tiles
// ---------------------------------------------------------------------------------
// NW = 8: at least 4 waves per SIMD (two 8-wave workgroups per CU) -- left to itself the compiler takes 130 registers and
// only one workgroup fits
template <typename T, int NW = 4, bool DROP = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void attn_bwd_dkv_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* qimg = smem;                                   // Q~ rows
    unsigned char* doimg = smem + AT<T>::ROW_IMG;                 // dO rows
    unsigned char* qtr = smem + 2 * AT<T>::ROW_IMG;               // Q~ transposed-read image
    unsigned char* dotr = qtr + AT<T>::TR_IMG;                    // dO transposed-read image
    float* lse_s = reinterpret_cast<float*>(dotr + AT<T>::TR_IMG);
    float* del_s = lse_s + BT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int ktile = blockIdx.x % p.qtiles;  // same tiling for keys
    const int bh = blockIdx.x / p.qtiles;
    const int head = bh % p.heads, b = bh / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.N * C + head * HD;
    const float scale = 0.125f;
    const int64_t statbase = ((int64_t)b * p.heads + head) * p.N;

    __shared__ float fl[64];
    if (threadIdx.x < 64) fl[threadIdx.x] = 0.f;
    const int key = ktile * (16 * NW) + wave * 16 + s;
    uint4 kf[AT<T>::NKK], vf[AT<T>::NKK];
    load_row_frag<T, true>(kf, kb, ld, key, p.N, p.E, g, p.cos_tab, p.heads, head, 1.0f);
    load_row_frag<T, false>(vf, vb, ld, key, p.N, p.E, g, nullptr, p.heads, head, 1.0f);

    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dk[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const int nqt = (p.N + BT - 1) / BT;
    TileFetch<T, true, 64 * NW> fq;   // the next query tile (and its statistics) travel in registers while this one is multiplied
    TileFetch<T, false, 64 * NW> fdo;
    const int stl = threadIdx.x & (BT - 1);  // every thread fetches a statistic (unconditional load); the first 64 commit
    float l_n, d_n;
    fq.fetch(qb, ld, 0, p.N, p.E, p.cos_tab, p.heads, head);
    fdo.fetch(dob, C, 0, p.N, p.E, nullptr, p.heads, head);
    l_n = p.lse[statbase + min(stl, p.N - 1)];
    d_n = p.delta[statbase + min(stl, p.N - 1)];
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        fq.template commit<true, true>(qimg, qtr, qt * BT, p.N, p.E, scale);
        fdo.template commit<true, true>(doimg, dotr, qt * BT, p.N, p.E, 1.0f);
        if (threadIdx.x < BT) {
            const int qq = qt * BT + threadIdx.x;
            lse_s[threadIdx.x] = RowExp<T>::prep(qq < p.N ? l_n : 0.f);
            del_s[threadIdx.x] = qq < p.N ? d_n : 0.f;
        }
        const int nxt = min(qt + 1, nqt - 1) * BT;
        fq.fetch(qb, ld, nxt, p.N, p.E, p.cos_tab, p.heads, head);
        fdo.fetch(dob, C, nxt, p.N, p.E, nullptr, p.heads, head);
        l_n = p.lse[statbase + min(nxt + stl, p.N - 1)];
        d_n = p.delta[statbase + min(nxt + stl, p.N - 1)];
        __syncthreads();
        // no masks (see the dq kernel): padding queries have zero rows in the four images and finite statistics
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // half steps of 32 queries
            f32x4_t sacc[4], dpacc[4];
            rows_times_frag_n<T, 2>(sacc, qimg + h * 32 * AT<T>::ROWB, s, g, kf);     // S[q = 16t + 4g + r][key = s]
            PHASE_FENCE();
            rows_times_frag_n<T, 2>(dpacc, doimg + h * 32 * AT<T>::ROWB, s, g, vf);   // dP[q][key]
            PHASE_FENCE();
            float pr[4][4], ds[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 2) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + h * 32 + t * 16 + 4 * g);
                    const float4 d4 = *reinterpret_cast<const float4*>(del_s + h * 32 + t * 16 + 4 * g);
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pp = RowExp<T>::clamped(sacc[t][r], lv[r]);
                        float keepf = 1.0f;
                        if constexpr (DROP) {  // this lane's key, the tile's queries: one byte per (query, key)
                            const int qq = qt * BT + h * 32 + t * 16 + 4 * g + r;
                            keepf = p.amask[(((int64_t)b * p.heads + head) * p.N + min(qq, p.N - 1)) * p.Np + min(key, p.N - 1)] ? p.a_inv_keep : 0.f;
                        }
                        pr[t][r] = pp * keepf;
                        ds[t][r] = pp * (dpacc[t][r] * keepf - dvv[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[t][r] = ds[t][r] = 0.f;
                }
            }
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dv, dotr + h * 32 * AT<T>::TRB, s, g, pr);  // dV^T[d][key] += dO^T . P
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dk, qtr + h * 32 * AT<T>::TRB, s, g, ds);   // dK~^T[d][key] += Q~^T . dS
            PHASE_FENCE();
        }
    }
    dkv_epilogue<T>(p, dk, dv, kb, ld, C, b, head, key, s, g, fl);
    if (p.E < p.N) freq_flush<true>(fl, p.fpart);
}

// ---------------------------------------------------------------------------------
// "Resident" variants for short sequences (N <= 256, bf16): one workgroup per (batch, head) stages
// the whole K~/V (forward, dq) or Q~/dO (dk/dv) of the head in LDS ONCE, then its waves walk their
// 16-row tiles with no barrier in the loop.  Against the tiled kernels above this removes the
// per-q-tile re-staging of K/V (4x at N = 199) and every in-loop __syncthreads.
// ---------------------------------------------------------------------------------
template <typename T, bool COS, bool ROWIMG, bool TRIMG>
__device__ __forceinline__ void stage_all(unsigned char* rowimg, unsigned char* trimg, const T* __restrict__ base, int64_t ld, int nrows_pad, int N, int E,
                                          const float* __restrict__ cos_tab, int heads, int head, float scale) {
    constexpr int NCH = AT<T>::NCH;
    constexpr int EPV = AT<T>::EPV;
    constexpr int UN = 4;  // chunks in flight per thread (N <= 256: the whole operand in one batch of 512 threads)
    const int total = nrows_pad * NCH;
    for (int i0 = threadIdx.x; i0 < total; i0 += UN * blockDim.x) {
        Chunk<T, COS> ch[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = min(i0 + u * (int)blockDim.x, total - 1);
            ch[u].fetch(base, ld, i / NCH, N, E, (i % NCH) * EPV, cos_tab, heads, head);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < total) {
                const int r = i / NCH, c = i % NCH;
                const uint4 v = ch[u].value(r, N, E, scale);
                if constexpr (ROWIMG) st16(rowimg + r * AT<T>::ROWB + ((c ^ (r & 7)) << 4), v);
                if constexpr (TRIMG) st16(trimg + r * AT<T>::TRB + c * 16, v);
            }
        }
    }
}

template <typename T, int NW = 8>  // NW = 4: sequences of at most 64 tokens (4 tiles of 16: half of an 8-wave workgroup would idle)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_res_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nkt = (p.N + BT - 1) / BT;
    const int npad = nkt * BT;
    unsigned char* kimg = smem;
    unsigned char* vimg = smem + npad * AT<T>::ROWB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int head = blockIdx.x % p.heads, b = blockIdx.x / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const float scale = 0.125f;
    stage_all<T, true, true, false>(kimg, nullptr, kb, ld, npad, p.N, p.E, p.cos_tab, p.heads, head, 1.0f);
    stage_all<T, false, false, true>(nullptr, vimg, vb, ld, npad, p.N, p.E, nullptr, p.heads, head, 1.0f);
    __syncthreads();
    const int nq16 = (p.N + 15) / 16;
    for (int qt = wave; qt < nq16; qt += NW) {
        const int q = qt * 16 + s;
        uint4 qf[AT<T>::NKK];
        load_row_frag<T, true>(qf, qb, ld, q, p.N, p.E, g, p.cos_tab, p.heads, head, scale);
        f32x4_t oacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        // NT live 16-key groups, MASK = the tile holds padding keys (only the last tile of a sequence can)
        auto kstep = [&](const int kt, auto ntc, auto maskc) __attribute__((always_inline)) {
            constexpr int NT = decltype(ntc)::value;
            constexpr bool MASK = decltype(maskc)::value != 0;
            f32x4_t sacc[4];
            rows_times_frag_n<T, NT>(sacc, kimg + kt * BT * AT<T>::ROWB, s, g, qf);
            float pv[4][4];
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = sacc[t][r];
                    if constexpr (MASK) v = kt * BT + t * 16 + 4 * g + r < p.N ? v : -INFINITY;
                    pv[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = group_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float mp = RowExp<T>::prep(m_new);
            const float alpha = RowExp<T>::sub(m_run, mp);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = t < NT ? RowExp<T>::sub(pv[t][r], mp) : 0.f;
                    pv[t][r] = e;
                    psum += e;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
            imgT_times_regs_n<T, NT>(oacc, vimg + kt * BT * AT<T>::TRB, s, g, pv);
        };
        const int nf = p.N / BT;  // tiles of 64 real keys
        for (int kt = 0; kt < nf; ++kt) kstep(kt, IC<4>{}, IC<0>{});
        if (nf < nkt) {
            if (p.N - nf * BT > 32) kstep(nf, IC<4>{}, IC<1>{});
            else kstep(nf, IC<2>{}, IC<1>{});
        }
        const float l_tot = group_sum(l_run);
        const float inv = 1.0f / l_tot;
        {
            float ov[4][4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[dt][r] = oacc[dt][r] * inv;
            store_row64<T>(reinterpret_cast<T*>(p.o) + ((int64_t)b * p.N + min(q, p.N - 1)) * C + head * HD, ov, g, q < p.N);
            if (q < p.N && g == 0 && p.lse) p.lse[((int64_t)b * p.heads + head) * p.N + q] = m_run + logf(l_tot);
        }
    }
}

template <typename T, int NW = 8>  // NW = 4: sequences of at most 64 tokens (4 tiles of 16: half of an 8-wave workgroup would idle)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_dq_res_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nkt = (p.N + BT - 1) / BT;
    const int npad = (p.N + 31) & ~31;  // image rows: padding groups beyond it are never read (nt below)
    unsigned char* kimg = smem;         // padded-pitch images: row fragments AND transposed reads
    unsigned char* vimg = kimg + npad * AT<T>::TRB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int head = blockIdx.x % p.heads, b = blockIdx.x / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.N * C + head * HD;
    const T* ob = reinterpret_cast<const T*>(p.o) + (int64_t)b * p.N * C + head * HD;
    const float scale = 0.125f;
    __shared__ float fl[64];  // freqs-gradient partial of this (sample, head)
    if (threadIdx.x < 64) fl[threadIdx.x] = 0.f;
    stage_all<T, true, false, true>(nullptr, kimg, kb, ld, npad, p.N, p.E, p.cos_tab, p.heads, head, 1.0f);
    stage_all<T, false, false, true>(nullptr, vimg, vb, ld, npad, p.N, p.E, nullptr, p.heads, head, 1.0f);
    __syncthreads();
    const int nq16 = (p.N + 15) / 16;
    for (int qt = wave; qt < nq16; qt += NW) {
        const int q = qt * 16 + s;
        // every fetch of a q tile in one batch (unconditional, clamped rows): q~, dO and O fragments, the raw q values
        // and cos factors of the epilogue, the row's LSE -- instead of one memory round trip per operand
        const int qc = min(q, p.N - 1);
        uint4 qf[AT<T>::NKK], dof[AT<T>::NKK];
        Chunk<T, true> cq[AT<T>::NKK];
        Chunk<T, false> cdo[AT<T>::NKK], co[AT<T>::NKK];
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            const int d0 = (kk * 4 + g) * AT<T>::EPV;
            cq[kk].fetch(qb, ld, q, p.N, p.E, d0, p.cos_tab, p.heads, head);
            cdo[kk].fetch(dob, C, q, p.N, p.E, d0, nullptr, p.heads, head);
            co[kk].fetch(ob, C, q, p.N, p.E, d0, nullptr, p.heads, head);
        }
        const float lse_raw = p.lse[((int64_t)b * p.heads + head) * p.N + qc];
        const T* qraw = qb + (int64_t)qc * ld;
        const float* cpr = p.cos_tab + ((int64_t)max(qc - p.E, 0) * p.heads + head) * 32;
        float qr[4][4], cr[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int d0 = dt * 16 + 4 * g;
            if constexpr (sizeof(T) == 2) {
                const uint2 r = *reinterpret_cast<const uint2*>(qraw + d0);
                const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
#pragma unroll
                for (int j = 0; j < 4; ++j) qr[dt][j] = (float)h[j];
            } else {
                const float4 r = *reinterpret_cast<const float4*>(qraw + d0);
                qr[dt][0] = r.x; qr[dt][1] = r.y; qr[dt][2] = r.z; qr[dt][3] = r.w;
            }
            const float2 c2 = *reinterpret_cast<const float2*>(cpr + (d0 >> 1));
            cr[dt][0] = c2.x; cr[dt][1] = c2.y;
        }
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            qf[kk] = cq[kk].value(q, p.N, p.E, scale);
            dof[kk] = cdo[kk].value(q, p.N, p.E, 1.0f);
        }
        float dl = 0.f;
        {
#pragma unroll
            for (int kk = 0; kk < AT<T>::NKK; ++kk) {
                Vec16<T> a, bb;
                a.raw = dof[kk];
                bb.raw = co[kk].value(q, p.N, p.E, 1.0f);
#pragma unroll
                for (int j = 0; j < AT<T>::EPV; ++j) dl += a.get(j) * bb.get(j);
            }
        }
        const float delta = group_sum(dl);
        const float lse = RowExp<T>::prep(q < p.N ? lse_raw : 0.f);
        if (q < p.N && g == 0) p.delta[((int64_t)b * p.heads + head) * p.N + q] = delta;
        f32x4_t dq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // No masks in the loop: the image rows of the padding keys (N <= key < npad) are zero, so whatever finite dS they
        // get multiplies a zero row of K~; lanes of padding queries are not stored.  exp's argument is <= 0 for every real
        // (query, key) pair, the clamp only keeps the padding ones finite.
        // Steps of 32 keys (two 16-key groups, one MFMA contraction step of dS . K~): the images are staged in units of 32 rows,
        // so every step is the same straight-line body -- and half the live registers of a 64-key step, which this kernel
        // needs (128-register budget: two workgroups per CU).
        for (int k0 = 0; k0 < npad; k0 += 32) {
            f32x4_t sacc[4], dpacc[4];
            rows_times_frag_pad_n<T, 2>(sacc, kimg + k0 * AT<T>::TRB, s, g, qf);
            PHASE_FENCE();
            rows_times_frag_pad_n<T, 2>(dpacc, vimg + k0 * AT<T>::TRB, s, g, dof);
            PHASE_FENCE();
            float ds[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[t][r] = t < 2 ? RowExp<T>::clamped(sacc[t][r], lse) * (dpacc[t][r] - delta) : 0.f;
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dq, kimg + k0 * AT<T>::TRB, s, g, ds);
            PHASE_FENCE();
        }
        {
            // the d cos / d freqs table entries of this row: fetched here (after the key loop: held across it they would spill)
            const float* sxp = p.dsin + ((int64_t)max(qc - p.E, 0) * p.heads + head) * 32;
            const float* syp = sxp + (int64_t)(p.N - p.E) * p.heads * 32;
            float gp[4][2], dx[4][2], dy[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float2 a = *reinterpret_cast<const float2*>(sxp + ((dt * 16 + 4 * g) >> 1));
                const float2 c = *reinterpret_cast<const float2*>(syp + ((dt * 16 + 4 * g) >> 1));
                dx[dt][0] = a.x; dx[dt][1] = a.y;
                dy[dt][0] = c.x; dy[dt][1] = c.y;
            }
            const bool row = q < p.N, img = row && q >= p.E;
            T* dqp = reinterpret_cast<T*>(p.dqkv) + ((int64_t)b * p.N + qc) * ld + head * HD;
            float oq[4][4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const float c0 = img ? cr[dt][0] * scale : scale, c1 = img ? cr[dt][1] * scale : scale;
                oq[dt][0] = dq[dt][0] * c0; oq[dt][1] = dq[dt][1] * c0; oq[dt][2] = dq[dt][2] * c1; oq[dt][3] = dq[dt][3] * c1;
                gp[dt][0] = img ? scale * (dq[dt][0] * qr[dt][0] + dq[dt][1] * qr[dt][1]) : 0.f;
                gp[dt][1] = img ? scale * (dq[dt][2] * qr[dt][2] + dq[dt][3] * qr[dt][3]) : 0.f;
            }
            store_row64<T>(dqp, oq, g, row);
            if (p.E < p.N) freq_accum(fl, gp, dx, dy, s, g);
        }
    }
    if (p.E < p.N) freq_flush<false>(fl, p.fpart);
}

// Diagnostic build only (-DATT_STAMP, tools/build_stamp.sh): per-phase s_memtime sums of wave 0 of one workgroup
#ifdef ATT_STAMP
__device__ unsigned long long g_att_stamp[8];
#define ATT_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define ATT_T(i) do { } while (0)
#endif

template <typename T, int NW = 8>  // NW = 4: sequences of at most 64 tokens (4 tiles of 16: half of an 8-wave workgroup would idle)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_dkv_res_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef ATT_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    const int nqt = (p.N + BT - 1) / BT;
    const int npad = (p.N + 31) & ~31;
    unsigned char* qimg = smem;  // padded-pitch images: row fragments AND transposed reads
    unsigned char* doimg = qimg + npad * AT<T>::TRB;
    float* lse_s = reinterpret_cast<float*>(doimg + npad * AT<T>::TRB);
    float* del_s = lse_s + nqt * BT;  // statistics are indexed by every query slot of a 64-query tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int head = blockIdx.x % p.heads, b = blockIdx.x / p.heads;
    const int C = p.heads * HD;
    const int64_t ld = 3 * C;
    const T* qb = reinterpret_cast<const T*>(p.qkv) + (int64_t)b * p.N * ld + head * HD;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = reinterpret_cast<const T*>(p.d_o) + (int64_t)b * p.N * C + head * HD;
    const float scale = 0.125f;
    const int64_t statbase = ((int64_t)b * p.heads + head) * p.N;
    __shared__ float fl[64];  // freqs-gradient partial of this (sample, head)
    if (threadIdx.x < 64) fl[threadIdx.x] = 0.f;
    stage_all<T, true, false, true>(nullptr, qimg, qb, ld, npad, p.N, p.E, p.cos_tab, p.heads, head, scale);
    stage_all<T, false, false, true>(nullptr, doimg, dob, C, npad, p.N, p.E, nullptr, p.heads, head, 1.0f);
    ATT_T(0);
    for (int i = threadIdx.x; i < nqt * BT; i += blockDim.x) {  // unconditional loads (clamped), zero by select
        const float l = p.lse[statbase + min(i, p.N - 1)], d = p.delta[statbase + min(i, p.N - 1)];
        lse_s[i] = RowExp<T>::prep(i < p.N ? l : 0.f);
        del_s[i] = i < p.N ? d : 0.f;
    }
    __syncthreads();
    ATT_T(1);
    const int nk16 = (p.N + 15) / 16;
    for (int ktile = wave; ktile < nk16; ktile += NW) {
        const int key = ktile * 16 + s;
        const int kc = min(key, p.N - 1);
        // every fetch of this key tile in one batch (see the dq kernel).  (Requesting the first tile's fragments before the
        // staging loads was tried in round 3: the chunks held across the staging spill, 212 -> 290 us for the pair of kernels.)
        uint4 kf[AT<T>::NKK], vf[AT<T>::NKK];
        Chunk<T, true> ck[AT<T>::NKK];
        Chunk<T, false> cv[AT<T>::NKK];
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            const int d0 = (kk * 4 + g) * AT<T>::EPV;
            ck[kk].fetch(kb, ld, key, p.N, p.E, d0, p.cos_tab, p.heads, head);
            cv[kk].fetch(vb, ld, key, p.N, p.E, d0, nullptr, p.heads, head);
        }
#pragma unroll
        for (int kk = 0; kk < AT<T>::NKK; ++kk) {
            kf[kk] = ck[kk].value(key, p.N, p.E, 1.0f);
            vf[kk] = cv[kk].value(key, p.N, p.E, 1.0f);
        }
        f32x4_t dk[4], dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dk[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        ATT_T(2);
        // No masks in the loop (see the dq kernel): padding queries have zero rows in both images and finite statistics (0).
        // steps of 32 queries (see the dq kernel)
        for (int q0 = 0; q0 < npad; q0 += 32) {
            f32x4_t sacc[4], dpacc[4];
            rows_times_frag_pad_n<T, 2>(sacc, qimg + q0 * AT<T>::TRB, s, g, kf);
            PHASE_FENCE();
            rows_times_frag_pad_n<T, 2>(dpacc, doimg + q0 * AT<T>::TRB, s, g, vf);
            PHASE_FENCE();
            float pr[4][4], ds[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < 2) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0 + t * 16 + 4 * g);
                    const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0 + t * 16 + 4 * g);
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pp = RowExp<T>::clamped(sacc[t][r], lv[r]);
                        pr[t][r] = pp;
                        ds[t][r] = pp * (dpacc[t][r] - dvv[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[t][r] = ds[t][r] = 0.f;
                }
            }
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dv, doimg + q0 * AT<T>::TRB, s, g, pr);
            PHASE_FENCE();
            imgT_times_regs_n<T, 2>(dk, qimg + q0 * AT<T>::TRB, s, g, ds);
            PHASE_FENCE();
        }
        ATT_T(3);
        dkv_epilogue<T>(p, dk, dv, kb, ld, C, b, head, key, s, g, fl);
        ATT_T(4);
    }
    if (p.E < p.N) freq_flush<true>(fl, p.fpart);
#ifdef ATT_STAMP
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_att_stamp[i] = tsum[i];
#endif
}
#ifdef ATT_STAMP
extern "C" int lnx_dbg_attn_stamps(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_att_stamp), 64); }
#endif

// ---------------------------------------------------------------------------------
// cos table and its backward to the learnable freqs
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void rope_cos_entry(const float* __restrict__ freqs, int heads, int H, int W, float* __restrict__ out, float* __restrict__ dsin, int i) {
    const int total = H * W * heads * 32;
    if (i >= total) return;
    const int j = i & 31;
    const int h = (i >> 5) % heads;
    const int n = (i >> 5) / heads;
    const float tx = (float)(n % W), ty = (float)(n / W);
    // theta = t_x * f_x + t_y * f_y in fp32, two rounded products then a rounded sum (einsum + add,
    // rope_2d_mhsa.py:136-142) -- keep them un-fused so the angle matches the reference bit for bit
    const float ax = __fmul_rn(tx, freqs[h * 32 + j]);
    const float ay = __fmul_rn(ty, freqs[(heads + h) * 32 + j]);
    const float th = __fadd_rn(ax, ay);
    out[i] = cosf(th);
    if (dsin) {  // d cos(theta) / d freqs[a, h, j] = -t_a sin(theta): what the attention backward weights its pair gradients with
        const float ms = -sinf(th);
        dsin[i] = tx * ms;
        dsin[total + i] = ty * ms;
    }
}

__global__ __launch_bounds__(256) void rope_cos_kernel(const float* __restrict__ freqs, int heads, int H, int W, float* __restrict__ out,
                                                       float* __restrict__ dsin) {
    rope_cos_entry(freqs, heads, H, W, out, dsin, blockIdx.x * 256 + threadIdx.x);
}

// the tables of several blocks in one launch (every RoPE block of a plan has its own freqs): blockIdx.y picks the table
struct RopeTabBatch {
    lnx_rope_table t[LNX_ROPE_TABLES_MAX];
};
__global__ __launch_bounds__(256) void rope_cos_batch_kernel(const RopeTabBatch b) {
    const lnx_rope_table& t = b.t[blockIdx.y];
    rope_cos_entry(t.freqs, t.heads, t.H, t.W, t.cos_out, t.dsin_out, blockIdx.x * 256 + threadIdx.x);
}

// dfreqs[a, h, j] += sum over the workgroups of head h of their partial [a][j]   (fixed order: deterministic given the partials)
// one 1024-thread workgroup per head: 16 slices of the partial list x 64 entries, four loads in flight per thread, LDS tree
__global__ __launch_bounds__(1024) void rope_freqs_reduce_kernel(const float* __restrict__ fpart, int B, int heads, int per_bh, float* __restrict__ dfreqs) {
    __shared__ float red[16][64];
    const int h = blockIdx.x, t = threadIdx.x & 63, sl = threadIdx.x >> 6;  // t = a * 32 + j
    const int n = B * per_bh;  // partials of this head: (b, k) -> ((b * heads + h) * per_bh + k)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    auto at = [&](int i) -> float {
        const int ii = min(i, n - 1);
        const int b = ii / per_bh, k = ii - b * per_bh;
        const float v = fpart[(((int64_t)b * heads + h) * per_bh + k) * 64 + t];
        return i < n ? v : 0.f;
    };
    for (int i = sl; i < n; i += 64) {
        a0 += at(i);
        a1 += at(i + 16);
        a2 += at(i + 32);
        a3 += at(i + 48);
    }
    red[sl][t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += red[k][t];
        dfreqs[((t >> 5) * heads + h) * 32 + (t & 31)] += acc;
    }
}

// the same fold for several attention backward calls in ONE launch (lnx_attn_bwd_args.defer_freqs + lnx_attn_bwd_flush): blockIdx.y picks
// the call.  Every RoPE block owns its freqs, so a backward segment of a plan has one of these small folds per block.
struct FreqEntry {
    const float* fpart;
    float* dfreqs;
    int B, heads, per_bh, pad_;
};
struct FreqBatch {
    FreqEntry e[LNX_ATTN_DEFER_MAX];
};
__global__ __launch_bounds__(1024) void rope_freqs_reduce_batch_kernel(const FreqBatch fb) {
    __shared__ float red[16][64];
    const FreqEntry& q = fb.e[blockIdx.y];
    if ((int)blockIdx.x >= q.heads) return;  // (uniform per workgroup: the grid's width is the largest head count of the batch)
    const float* __restrict__ fpart = q.fpart;
    const int heads = q.heads, per_bh = q.per_bh;
    const int h = blockIdx.x, t = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int n = q.B * per_bh;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    auto at = [&](int i) -> float {
        const int ii = min(i, n - 1);
        const int b = ii / per_bh, k = ii - b * per_bh;
        const float v = fpart[(((int64_t)b * heads + h) * per_bh + k) * 64 + t];
        return i < n ? v : 0.f;
    };
    for (int i = sl; i < n; i += 64) {  // (the order of rope_freqs_reduce_kernel: same bits)
        a0 += at(i);
        a1 += at(i + 16);
        a2 += at(i + 32);
        a3 += at(i + 48);
    }
    red[sl][t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sl == 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += red[k][t];
        q.dfreqs[((t >> 5) * heads + h) * 32 + (t & 31)] += acc;
    }
}

// postponed folds of this thread's lnx_attn_bwd calls (raw pointers: the caller keeps partials and targets alive until the flush)
thread_local FreqEntry g_freq_pending[LNX_ATTN_DEFER_MAX];
thread_local int g_freq_n = 0;
thread_local hipStream_t g_freq_stream = nullptr;

int freq_flush(hipStream_t st) {
    if (g_freq_n == 0) return 0;
    LNX_CHECK(st == g_freq_stream, "lnx_attn_bwd_flush: the postponed folds belong to another stream");
    FreqBatch fb{};
    int most = 0;
    for (int i = 0; i < g_freq_n; ++i) {
        fb.e[i] = g_freq_pending[i];
        if (fb.e[i].heads > most) most = fb.e[i].heads;
    }
    hipLaunchKernelGGL(rope_freqs_reduce_batch_kernel, dim3(most, g_freq_n), dim3(1024), 0, st, fb);
    g_freq_n = 0;
    LNX_LAUNCH_CHECK();
    return 0;
}

template <typename K> void set_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int check_attn(int dtype, int B, int N, int E, int heads, const char* who) {
    LNX_CHECK(dtype == LNX_F32 || dtype == LNX_BF16, "%s: bad dtype %d", who, dtype);
    LNX_CHECK(B > 0 && N > 0 && heads > 0 && E >= 0 && E <= N, "%s: bad shape B=%d N=%d E=%d heads=%d", who, B, N, E, heads);
    return 0;
}

}  // namespace

extern "C" int lnx_rope_cos_table(const float* freqs, int heads, int H, int W, float* cos_out, float* dsin_out, void* stream) {
    LNX_CHECK(freqs && cos_out && heads > 0 && H > 0 && W > 0, "lnx_rope_cos_table: bad arguments");
    const int total = H * W * heads * 32;
    hipLaunchKernelGGL(rope_cos_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, freqs, heads, H, W, cos_out, dsin_out);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_rope_cos_tables(const lnx_rope_table* t, int n, void* stream) {
    LNX_CHECK(t && n > 0, "lnx_rope_cos_tables: bad arguments");
    for (int i = 0; i < n; ++i) LNX_CHECK(t[i].freqs && t[i].cos_out && t[i].heads > 0 && t[i].H > 0 && t[i].W > 0, "lnx_rope_cos_tables: bad table");
    for (int i0 = 0; i0 < n; i0 += LNX_ROPE_TABLES_MAX) {
        RopeTabBatch b{};
        const int m = n - i0 < LNX_ROPE_TABLES_MAX ? n - i0 : LNX_ROPE_TABLES_MAX;
        int most = 0;
        for (int i = 0; i < m; ++i) {
            b.t[i] = t[i0 + i];
            const int total = t[i0 + i].H * t[i0 + i].W * t[i0 + i].heads * 32;
            if (total > most) most = total;
        }
        hipLaunchKernelGGL(rope_cos_batch_kernel, dim3(cdiv(most, 256), m), dim3(256), 0, (hipStream_t)stream, b);
        LNX_LAUNCH_CHECK();
    }
    return 0;
}

// floats of lnx_attn_bwd's freqs-gradient workspace: one [2][32] partial per workgroup of its finest tiling
extern "C" int64_t lnx_attn_bwd_ws_floats(int B, int N, int heads) { return (int64_t)B * heads * cdiv(N, BT) * 64; }

// tiled bf16 kernels: 8 waves (128 rows) per workgroup for sequences beyond the resident kernels' reach, where every staged
// tile is then shared by twice the rows; 4 waves for short ones (more workgroups) and with LNX_ATTN_NW=4 (A/B switch)
static bool tiled_nw8(int N) {
    static const bool force4 = getenv("LNX_ATTN_NW") != nullptr && atoi(getenv("LNX_ATTN_NW")) == 4;
    return !force4 && N > 128;
}

extern "C" int lnx_attn_fwd(const lnx_attn_args* a, void* stream) {
    LNX_CHECK(a && a->qkv && a->o, "lnx_attn_fwd: null operand");
    if (check_attn(a->dtype, a->B, a->N, a->E, a->heads, "lnx_attn_fwd")) return 1;
    LNX_CHECK(a->E == a->N || a->cos_tab, "lnx_attn_fwd: cos table missing");
    AttnP p{};
    p.qkv = a->qkv; p.cos_tab = a->cos_tab; p.o = a->o; p.lse = a->lse;
    p.B = a->B; p.N = a->N; p.E = a->E; p.heads = a->heads;
    p.qtiles = cdiv(a->N, BT);
    const int grid = a->B * a->heads * p.qtiles;
    hipStream_t st = (hipStream_t)stream;
    if (a->drop_mask) {  // attention-probability dropout: the 64-row tiled kernels with the DROP code
        LNX_CHECK(a->drop_inv_keep >= 1.0f && (((uintptr_t)a->drop_mask) & 3) == 0, "lnx_attn_fwd: drop_inv_keep >= 1 and a 4-byte aligned mask");
        p.amask = a->drop_mask; p.a_inv_keep = a->drop_inv_keep; p.Np = p.qtiles * BT;
        if (a->dtype == LNX_BF16) {
            hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 4, true>), dim3(grid), dim3(256), AT<bf16_t>::ROW_IMG + AT<bf16_t>::TR_IMG, st, p);
        } else {
            hipLaunchKernelGGL((attn_fwd_kernel<float, 4, true>), dim3(grid), dim3(256), AT<float>::ROW_IMG + AT<float>::TR_IMG, st, p);
        }
        LNX_LAUNCH_CHECK();
        return 0;
    }
    if (a->dtype == LNX_BF16 && a->N <= 256 && getenv("LNX_ATTN_TILED") == nullptr) {
        typedef bf16_t T;
        const int npad = p.qtiles * BT;
        const size_t lds = (size_t)npad * (AT<T>::ROWB + AT<T>::TRB);
        static bool once = false;
        if (!once) {
            set_lds(attn_fwd_res_kernel<T>, 256 * (AT<T>::ROWB + AT<T>::TRB));
            once = true;
        }
        if (a->N <= 64) hipLaunchKernelGGL((attn_fwd_res_kernel<T, 4>), dim3(a->B * a->heads), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((attn_fwd_res_kernel<T>), dim3(a->B * a->heads), dim3(512), lds, st, p);
    } else if (a->dtype == LNX_BF16) {
        const size_t lds = AT<bf16_t>::ROW_IMG + AT<bf16_t>::TR_IMG;
        if (tiled_nw8(a->N)) {  // 128 queries per workgroup
            p.qtiles = cdiv(a->N, 128);
            hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 8>), dim3(a->B * a->heads * p.qtiles), dim3(512), lds, st, p);
        } else {
            hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(grid), dim3(256), lds, st, p);
        }
    } else {
        const size_t lds = AT<float>::ROW_IMG + AT<float>::TR_IMG;
        hipLaunchKernelGGL((attn_fwd_kernel<float>), dim3(grid), dim3(256), lds, st, p);
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_attn_bwd(const lnx_attn_bwd_args* a, void* stream) {
    LNX_CHECK(a && a->qkv && a->o && a->lse && a->d_o && a->dqkv && a->delta, "lnx_attn_bwd: null operand");
    if (check_attn(a->dtype, a->B, a->N, a->E, a->heads, "lnx_attn_bwd")) return 1;
    LNX_CHECK(a->E == a->N || (a->cos_tab && a->dsin_tab && a->freq_ws && a->dfreqs), "lnx_attn_bwd: cos / d-cos tables, freqs-gradient workspace or dfreqs missing");
    AttnP p{};
    p.qkv = a->qkv; p.cos_tab = a->cos_tab; p.o = const_cast<void*>(a->o); p.lse = const_cast<float*>(a->lse);
    p.d_o = a->d_o; p.dqkv = a->dqkv; p.fpart = a->freq_ws; p.dsin = a->dsin_tab; p.delta = a->delta;
    p.B = a->B; p.N = a->N; p.E = a->E; p.heads = a->heads;
    p.qtiles = cdiv(a->N, BT);
    const int grid = a->B * a->heads * p.qtiles;
    hipStream_t st = (hipStream_t)stream;
    // after the two kernels: the per-workgroup partials of the freqs gradient -> dfreqs (per_bh = workgroups per (sample, head))
    if (a->defer_freqs && a->E < a->N) {  // checked before anything is launched
        LNX_CHECK(g_freq_n == 0 || g_freq_stream == st, "lnx_attn_bwd: postponed freqs folds are pending on another stream (lnx_attn_bwd_flush them first)");
        LNX_CHECK(g_freq_n < LNX_ATTN_DEFER_MAX, "lnx_attn_bwd: %d postponed freqs folds are pending; call lnx_attn_bwd_flush", LNX_ATTN_DEFER_MAX);
    }
    auto reduce_freqs = [&](int per_bh) {
        if (a->E >= a->N) return;
        if (a->defer_freqs) {
            g_freq_pending[g_freq_n++] = FreqEntry{p.fpart, a->dfreqs, a->B, a->heads, per_bh, 0};
            g_freq_stream = st;
            return;
        }
        hipLaunchKernelGGL(rope_freqs_reduce_kernel, dim3(a->heads), dim3(1024), 0, st, p.fpart, a->B, a->heads, per_bh, a->dfreqs);
    };
    if (a->drop_mask) {
        LNX_CHECK(a->drop_inv_keep >= 1.0f && (((uintptr_t)a->drop_mask) & 3) == 0, "lnx_attn_bwd: drop_inv_keep >= 1 and a 4-byte aligned mask");
        p.amask = a->drop_mask; p.a_inv_keep = a->drop_inv_keep; p.Np = p.qtiles * BT;
        if (a->dtype == LNX_BF16) {
            typedef bf16_t T;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 4, true>), dim3(grid), dim3(256), 2 * AT<T>::ROW_IMG + AT<T>::TR_IMG, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 4, true>), dim3(grid), dim3(256), 2 * AT<T>::ROW_IMG + 2 * AT<T>::TR_IMG + 2 * BT * sizeof(float), st, p);
        } else {
            typedef float T;
            const size_t lds_k = 2 * AT<T>::ROW_IMG + 2 * AT<T>::TR_IMG + 2 * BT * sizeof(float);
            static bool once = false;
            if (!once) {
                set_lds(attn_bwd_dkv_kernel<T, 4, true>, lds_k);
                once = true;
            }
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 4, true>), dim3(grid), dim3(256), 2 * AT<T>::ROW_IMG + AT<T>::TR_IMG, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 4, true>), dim3(grid), dim3(256), lds_k, st, p);
        }
        reduce_freqs(p.qtiles);
        LNX_LAUNCH_CHECK();
        return 0;
    }
    if (a->dtype == LNX_BF16 && a->N <= 256 && getenv("LNX_ATTN_TILED") == nullptr) {
        typedef bf16_t T;
        const int npad = (a->N + 31) & ~31;
        const size_t lds_q = (size_t)npad * (2 * AT<T>::TRB);
        const size_t lds_k = (size_t)npad * (2 * AT<T>::TRB) + (size_t)p.qtiles * BT * 2 * sizeof(float);
        static bool once = false;
        if (!once) {
            set_lds(attn_bwd_dq_res_kernel<T>, 256 * (2 * AT<T>::TRB));
            set_lds(attn_bwd_dkv_res_kernel<T>, 256 * (2 * AT<T>::TRB + 2 * sizeof(float)));
            once = true;
        }
        if (a->N <= 64) {
            hipLaunchKernelGGL((attn_bwd_dq_res_kernel<T, 4>), dim3(a->B * a->heads), dim3(256), lds_q, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<T, 4>), dim3(a->B * a->heads), dim3(256), lds_k, st, p);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_res_kernel<T>), dim3(a->B * a->heads), dim3(512), lds_q, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_res_kernel<T>), dim3(a->B * a->heads), dim3(512), lds_k, st, p);
        }
        reduce_freqs(1);
    } else if (a->dtype == LNX_BF16) {
        typedef bf16_t T;
        const size_t lds_q = 2 * AT<T>::ROW_IMG + AT<T>::TR_IMG;
        const size_t lds_k = 2 * AT<T>::ROW_IMG + 2 * AT<T>::TR_IMG + 2 * BT * sizeof(float);
        if (tiled_nw8(a->N)) {  // 128 queries / keys per workgroup
            p.qtiles = cdiv(a->N, 128);
            const int g8 = a->B * a->heads * p.qtiles;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 8>), dim3(g8), dim3(512), lds_q, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 8>), dim3(g8), dim3(512), lds_k, st, p);
            reduce_freqs(p.qtiles);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), dim3(grid), dim3(256), lds_q, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), dim3(grid), dim3(256), lds_k, st, p);
            reduce_freqs(p.qtiles);
        }
    } else {
        typedef float T;
        const size_t lds_q = 2 * AT<T>::ROW_IMG + AT<T>::TR_IMG;
        const size_t lds_k = 2 * AT<T>::ROW_IMG + 2 * AT<T>::TR_IMG + 2 * BT * sizeof(float);
        static bool once = false;
        if (!once) {
            set_lds(attn_bwd_dkv_kernel<T>, lds_k);
            once = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), dim3(grid), dim3(256), lds_q, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), dim3(grid), dim3(256), lds_k, st, p);
        reduce_freqs(p.qtiles);
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_attn_bwd_flush(void* stream) { return freq_flush((hipStream_t)stream); }

extern "C" int lnx_attn_bwd_discard(void) {
    const int n = g_freq_n;
    g_freq_n = 0;
    return n;
}
