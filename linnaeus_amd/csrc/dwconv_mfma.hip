// Depthwise 7x7 convolution on the matrix cores (gfx950), used when the compute type is bf16.
// Reference: nn.Conv2d(dim, dim, 7, padding=3, groups=dim), blocks/convnext.py:56-58, run under autocast: both
// operands rounded to bf16, fp32 accumulation.
//
// The VALU kernels of dwconv.hip spend 49 FMAs per output element and are instruction-bound at ~1/3 of the HBM
// rate.  A depthwise convolution has no reduction over channels, but for ONE channel and ONE kernel row ky the sum
// over kx is a banded (Toeplitz) matrix applied to an input row:
//
//     y[h, w] = sum_ky  sum_k T_ky[w, k] * x[h + ky, k],     T_ky[w, k] = wt[ky, k - w]  (0 <= k - w < 7)
//
// i.e. D[w_out, row] += T_ky[w_out, w_in] . X_ky[w_in, row]: one v_mfma_f32_16x16x32_bf16 per (channel, ky) yields a
// 16-column x 16-row output patch.  Only 7 of 32 k are non-zero per output, but the matrix core is ~25x the VALU rate,
// so the arithmetic all but disappears (28 MFMAs per wave per tile) and the kernel is left with staging, i.e. HBM.
//
// Layout: the NHWC tile is transposed on its way into LDS to channel-planar bf16 ([c][row][col], a b32 = two adjacent
// columns), so the B operand (8 consecutive columns of one row of one channel) is one aligned ds_read_b128.  The
// Toeplitz A operands of a wave's 2 channels x 7 kernel rows live in registers for the whole kernel (56 VGPRs).
// The next tile's pixels are fetched global->registers before the MFMAs and committed to LDS after them
// (issue-early / write-late, as the VALU kernels).
//
// Weight gradient: dW[ky, kx] = sum_{r, k} x[r, k + kx] * dy[r - ky, k] is, per input row r, a 7x7 product with the
// dy columns as the reduction: D[kx, ky] += A[kx, k] . B[k, ky], A a sliding window of the x row (built from five
// dword LDS reads + v_alignbit for the odd shifts), B an aligned row of dy.  Two channels share one 16x16 MFMA
// (rows/cols 0..7 and 8..15; the off-diagonal quadrants are ignored).
#include "common.hpp"
#include "dwconv_mfma.hpp"
#include <stdlib.h>

namespace {

typedef uint32_t u32;
typedef __attribute__((ext_vector_type(4))) u32 u32x4;

// Raw buffer addressing: a scalar 64-bit base (the tile's first pixel) in the descriptor plus a 32-bit per-lane byte
// offset that is tile-invariant, instead of a 64-bit address computation per lane per access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base) {
    // the base is workgroup-uniform; say so, or a value that went through a VGPR costs a waterfall loop per access
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(a >> 32)), lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)a);  // (the builtin returns int: no sign extension into the high half)
    const uint64_t u = (uint64_t)hi << 32 | lo;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, 0x7fffffff, 0x00020000);
}
constexpr u32 OOB = 0x80000000u;  // >= num_records: loads return 0, stores are dropped
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, u32 voff, u32 soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (u32)__builtin_amdgcn_readfirstlane((int)soff), 0);  // soff is uniform
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void buf_st16(__amdgpu_buffer_rsrc_t r, u32 voff, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, r, voff, 0, 0);
}
constexpr int CPW = 2;  // channels per wave: 2 x 7 Toeplitz operands = 56 VGPRs, the kernels stay under 128

__device__ __forceinline__ u32 pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(u32, v);
}

// One staged tile: ROWS x 2*CPAIRS pixels x CBM channels, NTHR threads.  issue(): NHWC global -> registers (16-byte loads, two
// adjacent columns per unit; zero outside the image).  commit(): registers -> channel-planar bf16 LDS, plane c at
// c*PLANE bytes, row pitch PITCH bytes, one ds_write_b32 = the two columns of one channel.  The per-unit offsets are
// tile-invariant and computed once (init), so a tile costs one scalar base + 3 compares per unit.
template <typename T, int NTHR, int CBM, int ROWS, int CPAIRS, int PITCH, int PLANE>
struct Stage {
    static constexpr int CH = 16 / sizeof(T);  // channels per 16-byte load
    static constexpr int NCG = CBM / CH;
    static constexpr int NU = ROWS * CPAIRS * NCG;
    static constexpr int NPU = (NU + NTHR - 1) / NTHR;
    uint4 p0[NPU], p1[NPU];
    u32 goff[NPU];  // byte offset of the unit's first pixel from the tile's first staged pixel (unsigned: scalar base + 32-bit lane offset addressing)
    int rc[NPU];    // row | first column << 8; -1: idle unit of the last round
    int cgoff;      // LDS byte offset of this thread's first channel plane (the same for all its units: NTHR % NCG == 0)
    static_assert(NTHR % NCG == 0, "one channel group per thread");

    __device__ __forceinline__ void init(int W, int C) {
#pragma unroll
        for (int k = 0; k < NPU; ++k) {
            const int u = threadIdx.x + NTHR * k;
            const int cg = u % NCG, cp = (u / NCG) % CPAIRS, row = u / (NCG * CPAIRS);
            const bool on = u < NU;
            goff[k] = on ? (u32)((row * W + 2 * cp) * C + cg * CH) * (u32)sizeof(T) : 0u;
            cgoff = cg * CH * PLANE;
            rc[k] = on ? (row | (2 * cp) << 8) : -1;
        }
    }

    // (hs, ws): image coordinates of the tile's first staged pixel (may be negative); src already points at channel c0
    // (hs, ws): image coordinates of the tile's first staged pixel (may be negative); src already points at channel c0.
    // No branch around the loads: a lane outside the image gets an offset beyond the descriptor's range, for which
    // the buffer load returns zero without touching memory.  (A conditional load merged with a zero makes the
    // compiler wait for the load on the spot -- s_waitcnt vmcnt(0) -- which serialises every fetch.)  `on` false:
    // nothing is fetched at all, so the calls stay unconditional.
    struct Tile {
        __amdgpu_buffer_rsrc_t rs;
        int hs, ws;
        bool on;
    };
    static __device__ __forceinline__ Tile tile(const T* __restrict__ src, int b, int hs, int ws, int H, int W, int C, bool on) {
        Tile t;
        t.rs = tile_rsrc(src + (((int64_t)b * H + hs) * W + ws) * C);  // only in-image lanes dereference it
        t.hs = hs; t.ws = ws; t.on = on;
        return t;
    }
    static constexpr int NLOAD = 2 * NPU;
    // load number i (unit i / 2, column i % 2) of the tile
    __device__ __forceinline__ void issue_one(const Tile& t, int i, int H, int W, int C) {
        const int k = i >> 1;
        const int h = t.hs + (rc[k] & 255), w = t.ws + (rc[k] >> 8) + (i & 1);
        const bool in = t.on && (NTHR * (k + 1) <= NU || rc[k] >= 0) && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
#ifndef DW_NOLOAD  // diagnostic builds: tools/build_dw_ablate.sh
        const uint4 v = buf_ld16(t.rs, in ? goff[k] : OOB, (i & 1) ? (u32)C * (u32)sizeof(T) : 0u);
#else
        const uint4 v = make_uint4(0u, 0u, 0u, 0u);
#endif
        if (i & 1) p1[k] = v; else p0[k] = v;
    }
    __device__ __forceinline__ void issue(const T* __restrict__ src, int b, int hs, int ws, int H, int W, int C, bool on = true) {
        const Tile t = tile(src, b, hs, ws, H, W, C, on);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) issue_one(t, i, H, W, C);
    }

    __device__ __forceinline__ void commit(char* __restrict__ dst) const {
#pragma unroll
        for (int k = 0; k < NPU; ++k) {
            if (NTHR * (k + 1) <= NU || rc[k] >= 0) {
                char* d = dst + cgoff + (rc[k] & 255) * PITCH + (rc[k] >> 8) * 2;
                if constexpr (sizeof(T) == 4) {
                    const float* f0 = reinterpret_cast<const float*>(&p0[k]);
                    const float* f1 = reinterpret_cast<const float*>(&p1[k]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32*>(d + j * PLANE) = pack2(f0[j], f1[j]);
                } else {
                    const u32* a = reinterpret_cast<const u32*>(&p0[k]);
                    const u32* bq = reinterpret_cast<const u32*>(&p1[k]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<u32*>(d + (2 * j) * PLANE) = __builtin_amdgcn_perm(bq[j], a[j], 0x05040100u);
                        *reinterpret_cast<u32*>(d + (2 * j + 1) * PLANE) = __builtin_amdgcn_perm(bq[j], a[j], 0x07060302u);
                    }
                }
            }
        }
    }

    // per-channel sums of the staged values (bias gradient): s[j] += channel cg*CH + j
    __device__ __forceinline__ void add_channel_sums(float (&s)[CH]) const {
#pragma unroll
        for (int k = 0; k < NPU; ++k) {
            if constexpr (sizeof(T) == 4) {
                const float* f0 = reinterpret_cast<const float*>(&p0[k]);
                const float* f1 = reinterpret_cast<const float*>(&p1[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += f0[j] + f1[j];
            } else {
                const u32* a = reinterpret_cast<const u32*>(&p0[k]);
                const u32* bq = reinterpret_cast<const u32*>(&p1[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s[2 * j] += __builtin_bit_cast(float, a[j] << 16) + __builtin_bit_cast(float, bq[j] << 16);
                    s[2 * j + 1] += __builtin_bit_cast(float, a[j] & 0xffff0000u) + __builtin_bit_cast(float, bq[j] & 0xffff0000u);
                }
            }
        }
    }
};

// blockIdx -> (pair index, which half); the grid is 16 * ceil(pairs / 8)
__device__ __forceinline__ void pair_of(int& pair, int& half) {
    const int within = blockIdx.x & 15;
    pair = (blockIdx.x >> 4) * 8 + (within & 7);
    half = within >> 3;
}

template <int NTHR> __device__ __forceinline__ void zero_lds(char* base, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16) *reinterpret_cast<uint4*>(base + i) = make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ f32x4_t mfma32(const uint4& a, const uint4& b, f32x4_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// forward / data gradient: 16 waves x 2 channels = 32 channels (128 contiguous bytes of an fp32 pixel) per workgroup,
// one workgroup per CU walking a contiguous run of tiles (row-major, so the 6 columns shared with the previous tile
// are L2 hits).  The next tile's pixels are fetched into registers during the MFMA loop (one load per kernel row, so
// the texture-address unit works under the MFMAs) and written to LDS after the output tile has been stored.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FT = 1024, FCB = 32;
constexpr int MT = 14;                         // output tile edge (14 | 56, 28: no partial tiles in the conv stages)
constexpr int MI = MT + 6;                     // input rows / columns that carry data
constexpr int XROWS = MI + 2;                  // + 2 zero rows read by the idle MFMA columns n = 14, 15
constexpr int XPITCH = 48;                     // 24 bf16: columns 20..23 stay zero; k = 24..31 meets zero taps only
constexpr int XPLANE = XROWS * XPITCH + 16;    // +16: successive 4-channel groups land 16 banks apart on commit
constexpr int X_BYTES = FCB * XPLANE;
template <typename TY> struct OutT {
    static constexpr int PITCH = FCB * sizeof(TY) + (sizeof(TY) == 2 ? 8 : 16);  // bytes per pixel, padded against bank conflicts
    static constexpr int PARTS = FCB * sizeof(TY) / 16;                         // 16-byte store units per pixel
    static constexpr int NOU = MT * MT * PARTS;
    static constexpr int NPO = (NOU + FT - 1) / FT;
    static constexpr int BYTES = MT * MT * PITCH;
};

struct MfP {
    const void* x;
    const float* w49;
    const float* bias;
    const float* res;
    void* y;
    int B, H, W, C;
    int tiles_h, tiles_w, tiles_per_wg, ntile, chunks;
};

struct TileAt {
    int b, h0, w0;
};
// row-major tile cursor: two integer divisions once, additions per step (a division is ~40 scalar instructions, and
// 16 waves each doing four per tile showed up as 15 % of the tile time)
struct TileCursor {
    int b, h0, w0;
    __device__ __forceinline__ void step(int H, int W, int th, int tw) {
        w0 += tw;
        if (w0 >= W) {
            w0 = 0;
            h0 += th;
            if (h0 >= H) {
                h0 = 0;
                ++b;
            }
        }
    }
};
__device__ __forceinline__ TileAt tile_at(int t, int tiles_h, int tiles_w, int th, int tw) {
    TileAt q;
    const int t2 = t / tiles_w;
    q.w0 = (t % tiles_w) * tw;
    q.h0 = (t2 % tiles_h) * th;
    q.b = t2 / tiles_h;
    return q;
}

// Diagnostic build only (-DDW_STAMP, tools/build_stamp.sh): per-phase s_memtime sums of wave 0 of one workgroup
#ifdef DW_STAMP
__device__ unsigned long long g_dwm_stamp[8];
#define DWM_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define DWM_T(i) do { } while (0)
#endif

template <typename TX, typename TY, bool FLIP>
__global__ __launch_bounds__(FT) void dwconv7_mfma_kernel(const MfP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Stage<TX, FT, FCB, MI, MI / 2, XPITCH, XPLANE> St;
    typedef OutT<TY> Ot;
    char* xt = smem;
    char* ot = smem + X_BYTES;
    const int cb = blockIdx.x / p.chunks;
    const int t_begin = (blockIdx.x % p.chunks) * p.tiles_per_wg;
    const int t_end = min(p.ntile, t_begin + p.tiles_per_wg);
    const int c0 = cb * FCB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const TX* xg = reinterpret_cast<const TX*>(p.x) + c0;

    St sa;
    sa.init(p.W, p.C);
    TileCursor cq, cq2;  // tiles t and t + 1
    {
        const TileAt q = tile_at(t_begin, p.tiles_h, p.tiles_w, MT, MT);
        cq.b = q.b; cq.h0 = q.h0; cq.w0 = q.w0;
        cq2 = cq;
        sa.issue(xg, cq2.b, cq2.h0 - 3, cq2.w0 - 3, p.H, p.W, p.C, true);
        cq2.step(p.H, p.W, MT, MT);
    }
    zero_lds<FT>(xt, X_BYTES);
    // taps of this channel block -> LDS (in the output-tile area), then the Toeplitz operands of this wave's channels:
    // lane (m = n, g) holds T[m][8g .. 8g+7] = wt[ky][k - m]
    float* wl = reinterpret_cast<float*>(ot);
    for (int i = threadIdx.x; i < 49 * FCB; i += FT) wl[i] = p.w49[(int64_t)(i / FCB) * p.C + c0 + (i % FCB)];
    __syncthreads();
    uint4 tz[CPW][7];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            u32 wd[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kx = 8 * g + 2 * jj + e - n;
                    const int kc = min(max(kx, 0), 6);  // clamped index + select: no branch per tap
                    const int tap = FLIP ? 48 - (ky * 7 + kc) : ky * 7 + kc;
                    const float wv = wl[tap * FCB + CPW * wave + c];
                    v[e] = kx == kc ? wv : 0.f;
                }
                wd[jj] = pack2(v[0], v[1]);
            }
            tz[c][ky] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
    float bv[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) bv[c] = p.bias ? p.bias[c0 + CPW * wave + c] : 0.f;
    __syncthreads();  // the tap staging area is the output tile from here on
    sa.commit(xt);
    __syncthreads();

    // output units of this thread (16 bytes of one pixel): global byte offset from the tile's first pixel, LDS offset
    u32 og[Ot::NPO];
    int ol[Ot::NPO], orc[Ot::NPO];
#pragma unroll
    for (int k = 0; k < Ot::NPO; ++k) {
        const int u = threadIdx.x + FT * k;
        const int px = u / Ot::PARTS, part = u % Ot::PARTS;
        const bool on = u < Ot::NOU;
        og[k] = on ? (u32)(((px / MT) * p.W + px % MT) * p.C) * (u32)sizeof(TY) + 16u * part : 0u;
        ol[k] = on ? px * Ot::PITCH + 16 * part : 0;
        orc[k] = on ? (px / MT | (px % MT) << 8) : 0xffff;  // idle units fail the row test
    }
    // B operand of lane (n, g): 8 columns of row n + ky; k = 24..31 re-reads k = 16..23 (its taps are zero)
    const char* xb = xt + (CPW * wave) * XPLANE + n * XPITCH + 16 * min(g, 2);
    char* ow = ot + wave * CPW * (int)sizeof(TY);

#ifdef DW_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    for (int t = t_begin; t < t_end; ++t) {
        const TileCursor q = cq;
        const int64_t tile0 = (((int64_t)q.b * p.H + q.h0) * p.W + q.w0) * p.C + c0;
        u32 go[Ot::NPO];
        uint4 rv[Ot::NPO];
#pragma unroll
        for (int k = 0; k < Ot::NPO; ++k) go[k] = q.h0 + (orc[k] & 255) < p.H && q.w0 + (orc[k] >> 8) < p.W ? og[k] : OOB;
        if constexpr (sizeof(TY) == 4) {  // residual of this tile (data gradient), fetched early
            if (p.res) {
                const __amdgpu_buffer_rsrc_t rr = tile_rsrc(p.res + tile0);
#pragma unroll
                for (int k = 0; k < Ot::NPO; ++k) rv[k] = buf_ld16(rr, go[k], 0u);
            }
        }
        // the fetch of tile t + 1 is spread over the MFMA loop: issued in one burst, 16 waves x 4 loads queue up in
        // the texture-address unit and every wave sits in that queue before its first MFMA
        const typename St::Tile ft = St::tile(xg, cq2.b, cq2.h0 - 3, cq2.w0 - 3, p.H, p.W, p.C, t + 1 < t_end);
        cq.step(p.H, p.W, MT, MT);
        cq2.step(p.H, p.W, MT, MT);
        DWM_T(0);

        f32x4_t acc[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // B operands AHEAD kernel rows ahead of their MFMAs; the fences pin that order, otherwise every MFMA waits
        // out the LDS latency of its own operand
        constexpr int AHEAD = sizeof(TY) == 2 ? 2 : 1;  // the fp32-output variants have no registers for a deeper ring
        uint4 bf[AHEAD + 1][CPW];
#pragma unroll
        for (int ky = 0; ky < AHEAD; ++ky)
#pragma unroll
            for (int c = 0; c < CPW; ++c) bf[ky][c] = *reinterpret_cast<const uint4*>(xb + c * XPLANE + ky * XPITCH);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (ky + AHEAD < 7) {
#pragma unroll
                for (int c = 0; c < CPW; ++c) bf[(ky + AHEAD) % (AHEAD + 1)][c] = *reinterpret_cast<const uint4*>(xb + c * XPLANE + (ky + AHEAD) * XPITCH);
            }
            if (ky < St::NLOAD) sa.issue_one(ft, ky, p.H, p.W, p.C);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CPW; ++c) acc[c] = mfma32(tz[c][ky], bf[ky % (AHEAD + 1)][c], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(St::NLOAD <= 7, "one load per kernel row");
        DWM_T(1);
        // D[m = 4g + i][n]: output column m of output row n, this wave's channels -> pixel-major output tile
        if (n < MT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * g + i;
                if (m < MT) {
                    char* o = ow + (n * MT + m) * Ot::PITCH;
                    if constexpr (sizeof(TY) == 2)
                        *reinterpret_cast<u32*>(o) = pack2(acc[0][i] + bv[0], acc[1][i] + bv[1]);
                    else
                        *reinterpret_cast<float2*>(o) = make_float2(acc[0][i] + bv[0], acc[1][i] + bv[1]);
                }
            }
        }
        DWM_T(2);
        __syncthreads();  // output tile complete; every wave is done reading the input tile
        DWM_T(3);
        {
            const __amdgpu_buffer_rsrc_t yr = tile_rsrc(reinterpret_cast<const TY*>(p.y) + tile0);
#pragma unroll
            for (int k = 0; k < Ot::NPO; ++k) {
                const char* o = ot + ol[k];
                if constexpr (sizeof(TY) == 2) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(o), hi = *reinterpret_cast<const uint2*>(o + 8);
#ifndef DW_NOSTORE
                    buf_st16(yr, go[k], make_uint4(lo.x, lo.y, hi.x, hi.y));
#else
                    if (lo.x == 0x12345678u) buf_st16(yr, go[k], make_uint4(lo.x, lo.y, hi.x, hi.y));
#endif
                } else {
                    float4 v = *reinterpret_cast<const float4*>(o);
                    if (p.res) {
                        v.x += __builtin_bit_cast(float, rv[k].x); v.y += __builtin_bit_cast(float, rv[k].y);
                        v.z += __builtin_bit_cast(float, rv[k].z); v.w += __builtin_bit_cast(float, rv[k].w);
                    }
                    buf_st16(yr, go[k], make_uint4(__builtin_bit_cast(u32, v.x), __builtin_bit_cast(u32, v.y), __builtin_bit_cast(u32, v.z), __builtin_bit_cast(u32, v.w)));
                }
            }
        }
        DWM_T(4);
        if (t + 1 < t_end) sa.commit(xt);
        DWM_T(5);
        __syncthreads();
        DWM_T(6);
    }
#ifdef DW_STAMP
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_dwm_stamp[i] = tsum[i];
#endif
}
#ifdef DW_STAMP
extern "C" int lnx_dbg_dwconv_mfma_stamps(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dwm_stamp), 64); }
#endif

// ------------------------------------------------------------------------------------------------------------------
// weight / bias gradient
// ------------------------------------------------------------------------------------------------------------------
// A workgroup is 8 waves x 2 channels = 16 channels; two of them fit a CU (<= 128 VGPRs, < 80 KB LDS), so one
// workgroup's fetch / transpose phases run under the other's MFMAs.  The two 16-channel halves of a 32-channel group
// are siblings: 8 apart in blockIdx = same XCD, dispatched back to back (pair_of), so the 128-byte lines they share
// are fetched into that XCD's L2 once.
constexpr int WT = 512, WCB = 16;
constexpr int WH = 14, WW = 28;                 // dy tile
constexpr int WXR = WH + 6, WXC = WW + 6;       // x tile
constexpr int WX_PITCH = 80;                    // 40 bf16: windows reach column 8*3 + 6 + 9
constexpr int WX_PLANE = WXR * WX_PITCH + 16;
constexpr int WX_BYTES = WCB * WX_PLANE;
constexpr int WD_ROWS = WXR + 7;                // dy rows -7 .. 19 (only 0..13 carry data, the rest stay zero)
constexpr int WD_PITCH = 64;                    // 32 bf16: K = 32 columns, 28..31 stay zero
constexpr int WD_PLANE = WD_ROWS * WD_PITCH + 16;
constexpr int WD_BYTES = WCB * WD_PLANE;

struct MwP {
    const void* x;
    const void* dy;
    float* dw;
    float* db;
    int B, H, W, C;
    int tiles_h, tiles_w, nwalk, per, ntile;  // walkers per 32-channel group, tiles per walker
};

template <typename TX, typename TDY>
__global__ __launch_bounds__(WT, 4) void dwconv7_mfma_wgrad_kernel(const MwP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xt = smem;
    char* dt = smem + WX_BYTES;
    float* dbl = reinterpret_cast<float*>(smem + WX_BYTES + WD_BYTES);
    typedef Stage<TX, WT, WCB, WXR, WXC / 2, WX_PITCH, WX_PLANE> Sx;
    typedef Stage<TDY, WT, WCB, WH, WW / 2, WD_PITCH, WD_PLANE> Sd;
    int pair, half;
    pair_of(pair, half);
    const int groups = p.C / (2 * WCB);
    if (pair >= p.nwalk * groups) return;  // grid padding
    const int walker = pair / groups;
    const int t_begin = walker * p.per, t_end = min(p.ntile, t_begin + p.per);  // a contiguous run of tiles (row-major)
    const int c0 = (2 * (pair % groups) + half) * WCB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const TX* xg = reinterpret_cast<const TX*>(p.x) + c0;
    const TDY* dg = reinterpret_cast<const TDY*>(p.dy) + c0;

    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float dbs[Sd::CH];
#pragma unroll
    for (int j = 0; j < Sd::CH; ++j) dbs[j] = 0.f;

    Sx sx;
    Sd sd;
    sx.init(p.W, p.C);
    sd.init(p.W, p.C);
    TileCursor cq;  // the tile being fetched
    {
        const TileAt q = tile_at(t_begin, p.tiles_h, p.tiles_w, WH, WW);
        cq.b = q.b; cq.h0 = q.h0; cq.w0 = q.w0;
        sx.issue(xg, q.b, q.h0 - 3, q.w0 - 3, p.H, p.W, p.C, t_begin < t_end);
        sd.issue(dg, q.b, q.h0, q.w0, p.H, p.W, p.C, t_begin < t_end);
        cq.step(p.H, p.W, WH, WW);
    }
    zero_lds<WT>(smem, WX_BYTES + WD_BYTES);
    if (threadIdx.x < WCB) dbl[threadIdx.x] = 0.f;
    __syncthreads();
    if (t_begin < t_end) {
        sx.commit(xt);
        sd.commit(dt + 7 * WD_PITCH);
        sd.add_channel_sums(dbs);
    }
    __syncthreads();

    // A (x windows): lane m = n: channel (m >> 3) of the pair, kx = m & 7; B (dy rows): channel (n >> 3), ky = n & 7
    const int kk = n & 7, ch = n >> 3;
    const u32 shift = (kk & 1) * 16;
    const char* xa0 = xt + (CPW * wave + ch) * WX_PLANE + 16 * g + 2 * (kk & ~1);
    const char* db0 = dt + (CPW * wave + ch) * WD_PLANE + (7 - kk) * WD_PITCH + 16 * g;
    constexpr int NL = Sx::NLOAD + Sd::NLOAD;
    static_assert(2 * NL <= WXR, "one fetch every other input row");

    for (int t = t_begin; t < t_end; ++t) {
        const bool more = t + 1 < t_end;
        // the next tile's fetches are spread over the MFMA loop (one every other input row); operands are read one row
        // ahead of their MFMA, the fences keep that order
        const typename Sx::Tile tx = Sx::tile(xg, cq.b, cq.h0 - 3, cq.w0 - 3, p.H, p.W, p.C, more);
        const typename Sd::Tile td = Sd::tile(dg, cq.b, cq.h0, cq.w0, p.H, p.W, p.C, more);
        cq.step(p.H, p.W, WH, WW);
        u32 da[2][5];
        uint4 bq[2];
        auto operands = [&](int r) {
            const u32* xa = reinterpret_cast<const u32*>(xa0 + r * WX_PITCH);
#pragma unroll
            for (int j = 0; j < 5; ++j) da[r & 1][j] = xa[j];
            bq[r & 1] = *reinterpret_cast<const uint4*>(db0 + r * WD_PITCH);
        };
        operands(0);
#pragma unroll
        for (int r = 0; r < WXR; ++r) {
            if (r + 1 < WXR) operands(r + 1);
            if ((r & 1) == 0 && r / 2 < NL) {
                if (r / 2 < Sx::NLOAD) sx.issue_one(tx, r / 2, p.H, p.W, p.C);
                else sd.issue_one(td, r / 2 - Sx::NLOAD, p.H, p.W, p.C);
            }
            __builtin_amdgcn_sched_barrier(0);
            const u32* d = da[r & 1];
            const uint4 a = make_uint4(__builtin_amdgcn_alignbit(d[1], d[0], shift), __builtin_amdgcn_alignbit(d[2], d[1], shift),
                                       __builtin_amdgcn_alignbit(d[3], d[2], shift), __builtin_amdgcn_alignbit(d[4], d[3], shift));
            acc = mfma32(a, bq[r & 1], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (more) {
            sx.commit(xt);
            sd.commit(dt + 7 * WD_PITCH);
            sd.add_channel_sums(dbs);
            __syncthreads();
        }
    }
    // D[m = 4g + i][n]: valid where both indices name the same channel of the pair and kx, ky < 7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = 4 * g + i;
        const int kx = m & 7;
        if ((m >> 3) == ch && kx < 7 && kk < 7) atomicAdd(p.dw + (int64_t)(c0 + CPW * wave + ch) * 49 + kk * 7 + kx, acc[i]);
    }
    if (p.db) {
        const int cg = threadIdx.x % Sd::NCG;
#pragma unroll
        for (int j = 0; j < Sd::CH; ++j) atomicAdd(&dbl[cg * Sd::CH + j], dbs[j]);
        __syncthreads();
        if (threadIdx.x < WCB) atomicAdd(p.db + c0 + threadIdx.x, dbl[threadIdx.x]);
    }
}

template <typename K> int set_lds(K kernel, int bytes) {
    LNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

template <typename TX, typename TY, bool FLIP> int launch_fwd(const MfP& p, int grid, hipStream_t st) {
    const int lds = X_BYTES + OutT<TY>::BYTES;
    static bool ready = false;  // one attribute call per instantiation
    if (!ready) {
        if (int rc = set_lds(dwconv7_mfma_kernel<TX, TY, FLIP>, lds)) return rc;
        ready = true;
    }
    hipLaunchKernelGGL((dwconv7_mfma_kernel<TX, TY, FLIP>), dim3(grid), dim3(FT), lds, st, p);
    return 0;
}

template <typename TX, typename TDY> int launch_wgrad(const MwP& p, int grid, hipStream_t st) {
    const int lds = WX_BYTES + WD_BYTES + WCB * (int)sizeof(float);
    static bool ready = false;
    if (!ready) {
        if (int rc = set_lds(dwconv7_mfma_wgrad_kernel<TX, TDY>, lds)) return rc;
        ready = true;
    }
    hipLaunchKernelGGL((dwconv7_mfma_wgrad_kernel<TX, TDY>), dim3(grid), dim3(WT), lds, st, p);
    return 0;
}

// CUs a launch of these one-workgroup-per-CU kernels may use: the device's minus the margin (LNX_CU_MARGIN / lnx_set_cu_margin).  Their
// tiles stay statically partitioned -- a workgroup walks a CONTIGUOUS run of tiles so that the 6 halo columns it shares with the
// previous tile are L2 hits, which a drawn order would give up; beside a resident collective kernel the cost was measured
// (profiles/r04_cu_hog.log: the whole step loses 3 % with 32-96 CUs held for the duration of every bucket's all-reduce).
int cus() {
    const int n = device_cus();
    return persistent_cus(n > 0 ? n : 256);
}

}  // namespace

bool lnx_dwconv_mfma_enabled() {
    static const bool on = getenv("LNX_DWCONV_VALU") == nullptr;  // A/B switch: the VALU kernels of dwconv.hip
    return on;
}

int lnx_dwconv7_mfma_fwd(const lnx_dwconv_args* a, hipStream_t st) {
    MfP p;
    p.x = a->x; p.w49 = a->w49; p.bias = a->bias; p.res = a->res; p.y = a->y;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C;
    const int code = a->x_dtype * 2 + a->y_dtype + (a->flip ? 4 : 0);
    p.tiles_h = cdiv(a->H, MT); p.tiles_w = cdiv(a->W, MT);
    const int64_t ntile = (int64_t)a->B * p.tiles_h * p.tiles_w;
    LNX_CHECK(ntile < (1ll << 31), "lnx_dwconv7_fwd: too many tiles");
    p.ntile = (int)ntile;
    // one resident workgroup per CU, each walks a contiguous run of tiles of its 32-channel block
    const int cblocks = a->C / FCB;
    int chunks = cus() / cblocks;
    if (chunks < 1) chunks = 1;
    if (chunks > p.ntile) chunks = p.ntile;
    p.tiles_per_wg = cdiv(p.ntile, chunks);
    p.chunks = cdiv(p.ntile, p.tiles_per_wg);
    const int grid = p.chunks * cblocks;
    LNX_CHECK(!(a->res && a->y_dtype != LNX_F32), "lnx_dwconv7_fwd: a residual needs an fp32 output");
    int rc = 0;
    switch (code) {
        case 1: rc = launch_fwd<float, bf16_t, false>(p, grid, st); break;
        case 2: rc = launch_fwd<bf16_t, float, false>(p, grid, st); break;
        case 3: rc = launch_fwd<bf16_t, bf16_t, false>(p, grid, st); break;
        case 5: rc = launch_fwd<float, bf16_t, true>(p, grid, st); break;
        case 6: rc = launch_fwd<bf16_t, float, true>(p, grid, st); break;
        case 7: rc = launch_fwd<bf16_t, bf16_t, true>(p, grid, st); break;
        default: LNX_CHECK(false, "lnx_dwconv7_fwd: the MFMA path needs a bf16 operand");
    }
    if (rc) return rc;
    LNX_LAUNCH_CHECK();
    return 0;
}

int lnx_dwconv7_mfma_wgrad(const lnx_dwconv_wgrad_args* a, hipStream_t st) {
    MwP p;
    p.x = a->x; p.dy = a->dy; p.dw = a->dw; p.db = a->db;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C;
    p.tiles_h = cdiv(a->H, WH); p.tiles_w = cdiv(a->W, WW);
    const int64_t ntile = (int64_t)a->B * p.tiles_h * p.tiles_w;
    LNX_CHECK(ntile < (1ll << 31), "lnx_dwconv7_wgrad: too many tiles");
    p.ntile = (int)ntile;
    const int groups = a->C / (2 * WCB);
    int walkers = cus() / groups;  // one resident sibling pair per CU
    if (walkers < 1) walkers = 1;
    if (walkers > p.ntile) walkers = p.ntile;
    p.per = cdiv(p.ntile, walkers);
    walkers = cdiv(p.ntile, p.per);  // same longest walk, no idle walkers
    p.nwalk = walkers;
    const int grid = 16 * cdiv(walkers * groups, 8);
    const int code = a->x_dtype * 2 + a->dy_dtype;
    int rc = 0;
    switch (code) {
        case 1: rc = launch_wgrad<float, bf16_t>(p, grid, st); break;
        case 2: rc = launch_wgrad<bf16_t, float>(p, grid, st); break;
        case 3: rc = launch_wgrad<bf16_t, bf16_t>(p, grid, st); break;
        default: LNX_CHECK(false, "lnx_dwconv7_wgrad: the MFMA path needs a bf16 operand");
    }
    if (rc) return rc;
    LNX_LAUNCH_CHECK();
    return 0;
}
