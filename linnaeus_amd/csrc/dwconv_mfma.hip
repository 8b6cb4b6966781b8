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
// Toeplitz A operands of a wave's 4 channels x 7 kernel rows live in registers for the whole kernel (112 VGPRs).
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
constexpr int NTHR = 512;  // 8 waves, 4 channels each
constexpr int CBM = 32;    // channels per workgroup

__device__ __forceinline__ u32 pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(u32, v);
}

// One staged tile: ROWS x 2*CPAIRS pixels x 32 channels.  issue(): NHWC global -> registers (16-byte loads, two
// adjacent columns per unit; zero outside the image).  commit(): registers -> channel-planar bf16 LDS, plane c at
// c*PLANE bytes, row pitch PITCH bytes, one ds_write_b32 = the two columns of one channel.
template <typename T, int ROWS, int CPAIRS, int PITCH, int PLANE>
struct Stage {
    static constexpr int CH = 16 / sizeof(T);  // channels per 16-byte load
    static constexpr int NCG = CBM / CH;
    static constexpr int NU = ROWS * CPAIRS * NCG;
    static constexpr int NPU = (NU + NTHR - 1) / NTHR;
    uint4 p0[NPU], p1[NPU];

    __device__ __forceinline__ void issue(const T* __restrict__ src, int b, int hs, int ws, int c0, int H, int W, int C) {
#pragma unroll
        for (int k = 0; k < NPU; ++k) {
            const int u = threadIdx.x + NTHR * k;
            p0[k] = make_uint4(0u, 0u, 0u, 0u);
            p1[k] = make_uint4(0u, 0u, 0u, 0u);
            if (u < NU) {
                const int cg = u % NCG, cp = (u / NCG) % CPAIRS, row = u / (NCG * CPAIRS);
                const int h = hs + row, w = ws + 2 * cp;
                if (h >= 0 && h < H) {
                    const int64_t off = (((int64_t)b * H + h) * W + w) * C + c0 + cg * CH;
                    if (w >= 0 && w < W) p0[k] = ld16(src + off);
                    if (w + 1 >= 0 && w + 1 < W) p1[k] = ld16(src + off + C);
                }
            }
        }
    }

    __device__ __forceinline__ void commit(char* __restrict__ dst) const {
#pragma unroll
        for (int k = 0; k < NPU; ++k) {
            const int u = threadIdx.x + NTHR * k;
            if (u < NU) {
                const int cg = u % NCG, cp = (u / NCG) % CPAIRS, row = u / (NCG * CPAIRS);
                char* d = dst + cg * CH * PLANE + row * PITCH + cp * 4;
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

__device__ __forceinline__ void zero_lds(char* base, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16) *reinterpret_cast<uint4*>(base + i) = make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ f32x4_t mfma32(const uint4& a, const uint4& b, f32x4_t acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------------
// forward / data gradient
// ------------------------------------------------------------------------------------------------------------------
constexpr int MT = 14;                         // output tile edge (14 | 56, 28: no partial tiles in the conv stages)
constexpr int MI = MT + 6;                     // input rows / columns that carry data
constexpr int XROWS = MI + 2;                  // + 2 zero rows read by the idle MFMA columns n = 14, 15
constexpr int XPITCH = 64;                     // 32 bf16: K = 32 input columns, 20..31 stay zero
constexpr int XPLANE = XROWS * XPITCH + 16;    // +16: successive 4-channel groups land 16 banks apart on commit
constexpr int X_BYTES = CBM * XPLANE;

template <typename TY> struct OutT {
    static constexpr int PITCH = CBM * sizeof(TY) + (sizeof(TY) == 2 ? 8 : 16);  // bytes per pixel, padded against bank conflicts
    static constexpr int PARTS = CBM * sizeof(TY) / 16;                         // 16-byte store units per pixel
    static constexpr int NOU = MT * MT * PARTS;
    static constexpr int NPO = (NOU + NTHR - 1) / NTHR;
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
__device__ __forceinline__ TileAt tile_at(int t, int tiles_h, int tiles_w, int th, int tw) {
    TileAt q;
    const int t2 = t / tiles_w;
    q.w0 = (t % tiles_w) * tw;
    q.h0 = (t2 % tiles_h) * th;
    q.b = t2 / tiles_h;
    return q;
}

template <typename TX, typename TY, bool FLIP>
__global__ __launch_bounds__(NTHR) void dwconv7_mfma_kernel(const MfP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xt = smem;
    char* ot = smem + X_BYTES;
    typedef Stage<TX, MI, MI / 2, XPITCH, XPLANE> St;
    typedef OutT<TY> Ot;
    const int cb = blockIdx.x / p.chunks;
    const int t_begin = (blockIdx.x % p.chunks) * p.tiles_per_wg;
    const int t_end = min(p.ntile, t_begin + p.tiles_per_wg);
    const int c0 = cb * CBM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const TX* xg = reinterpret_cast<const TX*>(p.x);

    St st;
    {
        const TileAt q = tile_at(t_begin, p.tiles_h, p.tiles_w, MT, MT);
        st.issue(xg, q.b, q.h0 - 3, q.w0 - 3, c0, p.H, p.W, p.C);
    }
    zero_lds(xt, X_BYTES);
    // taps of this channel block -> LDS (in the output-tile area), then the Toeplitz operands of this wave's channels
    float* wl = reinterpret_cast<float*>(ot);
    for (int i = threadIdx.x; i < 49 * CBM; i += NTHR) wl[i] = p.w49[(int64_t)(i / CBM) * p.C + c0 + (i % CBM)];
    __syncthreads();
    uint4 tz[4][7];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            u32 wd[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kx = 8 * g + 2 * jj + e - n;  // T[m = n][k = 8g + j] = wt[ky][k - m]
                    const int kc = min(max(kx, 0), 6);      // clamped index + select: no branch per tap
                    const int tap = FLIP ? 48 - (ky * 7 + kc) : ky * 7 + kc;
                    const float wv = wl[tap * CBM + 4 * wave + c];
                    v[e] = kx == kc ? wv : 0.f;
                }
                wd[jj] = pack2(v[0], v[1]);
            }
            tz[c][ky] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
    float bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = p.bias ? p.bias[c0 + 4 * wave + c] : 0.f;
    __syncthreads();  // the tap staging area is the output tile from here on
    st.commit(xt);
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const TileAt q = tile_at(t, p.tiles_h, p.tiles_w, MT, MT);
        const bool more = t + 1 < t_end;
        if (more) {
            const TileAt qn = tile_at(t + 1, p.tiles_h, p.tiles_w, MT, MT);
            st.issue(xg, qn.b, qn.h0 - 3, qn.w0 - 3, c0, p.H, p.W, p.C);
        }
        // residual of this tile (data gradient), fetched early
        float4 rv[Ot::NPO];
        if constexpr (sizeof(TY) == 4) {
            if (p.res) {
#pragma unroll
                for (int k = 0; k < Ot::NPO; ++k) {
                    const int u = threadIdx.x + NTHR * k;
                    const int px = u / Ot::PARTS, part = u % Ot::PARTS;
                    const int h = q.h0 + px / MT, w = q.w0 + px % MT;
                    rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (u < Ot::NOU && h < p.H && w < p.W)
                        rv[k] = *reinterpret_cast<const float4*>(p.res + (((int64_t)q.b * p.H + h) * p.W + w) * p.C + c0 + 4 * part);
                }
            }
        }

        f32x4_t acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* xb = xt + (4 * wave) * XPLANE + n * XPITCH + 16 * g;
        // B operands one kernel row ahead of their MFMAs; the fences keep the compiler from hoisting all 28 reads
        uint4 bf[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bf[0][c] = *reinterpret_cast<const uint4*>(xb + c * XPLANE);
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            if (ky < 6) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bf[(ky + 1) & 1][c] = *reinterpret_cast<const uint4*>(xb + c * XPLANE + (ky + 1) * XPITCH);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma32(tz[c][ky], bf[ky & 1][c], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // D[m = 4g + i][n]: output column m of output row n, this wave's 4 channels -> pixel-major output tile
        if (n < MT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 4 * g + i;
                if (m < MT) {
                    char* o = ot + (n * MT + m) * Ot::PITCH + wave * 4 * (int)sizeof(TY);
                    if constexpr (sizeof(TY) == 2)
                        *reinterpret_cast<uint2*>(o) = make_uint2(pack2(acc[0][i] + bv[0], acc[1][i] + bv[1]), pack2(acc[2][i] + bv[2], acc[3][i] + bv[3]));
                    else
                        *reinterpret_cast<float4*>(o) = make_float4(acc[0][i] + bv[0], acc[1][i] + bv[1], acc[2][i] + bv[2], acc[3][i] + bv[3]);
                }
            }
        }
        __syncthreads();  // output tile complete; every wave is done reading the input tile
#pragma unroll
        for (int k = 0; k < Ot::NPO; ++k) {
            const int u = threadIdx.x + NTHR * k;
            const int px = u / Ot::PARTS, part = u % Ot::PARTS;
            const int h = q.h0 + px / MT, w = q.w0 + px % MT;
            if (u < Ot::NOU && h < p.H && w < p.W) {
                const char* o = ot + px * Ot::PITCH + 16 * part;
                const int64_t off = (((int64_t)q.b * p.H + h) * p.W + w) * p.C + c0;
                if constexpr (sizeof(TY) == 2) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(o), hi = *reinterpret_cast<const uint2*>(o + 8);
                    st16(reinterpret_cast<bf16_t*>(p.y) + off + 8 * part, make_uint4(lo.x, lo.y, hi.x, hi.y));
                } else {
                    float4 v = *reinterpret_cast<const float4*>(o);
                    if (p.res) {
                        v.x += rv[k].x; v.y += rv[k].y; v.z += rv[k].z; v.w += rv[k].w;
                    }
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + off + 4 * part) = v;
                }
            }
        }
        if (more) st.commit(xt);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// weight / bias gradient
// ------------------------------------------------------------------------------------------------------------------
constexpr int WH = 14, WW = 28;                 // dy tile
constexpr int WXR = WH + 6, WXC = WW + 6;       // x tile
constexpr int WX_PITCH = 80;                    // 40 bf16: windows reach column 8*3 + 6 + 9
constexpr int WX_PLANE = WXR * WX_PITCH + 16;
constexpr int WX_BYTES = CBM * WX_PLANE;
constexpr int WD_ROWS = WXR + 7;                // dy rows -7 .. 19 (only 0..13 carry data, the rest stay zero)
constexpr int WD_PITCH = 64;                    // 32 bf16: K = 32 columns, 28..31 stay zero
constexpr int WD_PLANE = WD_ROWS * WD_PITCH + 16;
constexpr int WD_BYTES = CBM * WD_PLANE;

struct MwP {
    const void* x;
    const void* dy;
    float* dw;
    float* db;
    int B, H, W, C;
    int tiles_h, tiles_w, cblocks, ntile;
};

template <typename TX, typename TDY>
__global__ __launch_bounds__(NTHR) void dwconv7_mfma_wgrad_kernel(const MwP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xt = smem;
    char* dt = smem + WX_BYTES;
    float* dbl = reinterpret_cast<float*>(smem + WX_BYTES + WD_BYTES);
    typedef Stage<TX, WXR, WXC / 2, WX_PITCH, WX_PLANE> Sx;
    typedef Stage<TDY, WH, WW / 2, WD_PITCH, WD_PLANE> Sd;
    const int cb = blockIdx.x % p.cblocks;
    const int walker = blockIdx.x / p.cblocks;
    const int nwalk = gridDim.x / p.cblocks;
    const int c0 = cb * CBM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const TX* xg = reinterpret_cast<const TX*>(p.x);
    const TDY* dg = reinterpret_cast<const TDY*>(p.dy);

    f32x4_t acc[2];
    acc[0] = acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float dbs[Sd::CH];
#pragma unroll
    for (int j = 0; j < Sd::CH; ++j) dbs[j] = 0.f;

    Sx sx;
    Sd sd;
    if (walker < p.ntile) {
        const TileAt q = tile_at(walker, p.tiles_h, p.tiles_w, WH, WW);
        sx.issue(xg, q.b, q.h0 - 3, q.w0 - 3, c0, p.H, p.W, p.C);
        sd.issue(dg, q.b, q.h0, q.w0, c0, p.H, p.W, p.C);
    }
    zero_lds(smem, WX_BYTES + WD_BYTES);
    if (threadIdx.x < CBM) dbl[threadIdx.x] = 0.f;
    __syncthreads();
    if (walker < p.ntile) {
        sx.commit(xt);
        sd.commit(dt + 7 * WD_PITCH);
        sd.add_channel_sums(dbs);
    }
    __syncthreads();

    // A (x windows): lane m = n: channel (m >> 3) of the pair, kx = m & 7; B (dy rows): channel (n >> 3), ky = n & 7
    const int kk = n & 7, ch = n >> 3;
    const u32 shift = (kk & 1) * 16;
    const char* xa0 = xt + (4 * wave + ch) * WX_PLANE + 16 * g + 2 * (kk & ~1);
    const char* db0 = dt + (4 * wave + ch) * WD_PLANE + (7 - kk) * WD_PITCH + 16 * g;

    for (int t = walker; t < p.ntile; t += nwalk) {
        const bool more = t + nwalk < p.ntile;
        if (more) {
            const TileAt qn = tile_at(t + nwalk, p.tiles_h, p.tiles_w, WH, WW);
            sx.issue(xg, qn.b, qn.h0 - 3, qn.w0 - 3, c0, p.H, p.W, p.C);
            sd.issue(dg, qn.b, qn.h0, qn.w0, c0, p.H, p.W, p.C);
        }
#pragma unroll 4
        for (int r = 0; r < WXR; ++r)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const u32* xa = reinterpret_cast<const u32*>(xa0 + 2 * pr * WX_PLANE + r * WX_PITCH);
                const u32 d0 = xa[0], d1 = xa[1], d2 = xa[2], d3 = xa[3], d4 = xa[4];
                const uint4 a = make_uint4(__builtin_amdgcn_alignbit(d1, d0, shift), __builtin_amdgcn_alignbit(d2, d1, shift),
                                           __builtin_amdgcn_alignbit(d3, d2, shift), __builtin_amdgcn_alignbit(d4, d3, shift));
                const uint4 bq = *reinterpret_cast<const uint4*>(db0 + 2 * pr * WD_PLANE + r * WD_PITCH);
                acc[pr] = mfma32(a, bq, acc[pr]);
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
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * g + i;
            const int kx = m & 7;
            if ((m >> 3) == ch && kx < 7 && kk < 7) atomicAdd(p.dw + (int64_t)(c0 + 4 * wave + 2 * pr + ch) * 49 + kk * 7 + kx, acc[pr][i]);
        }
    if (p.db) {
        const int cg = threadIdx.x % Sd::NCG;
#pragma unroll
        for (int j = 0; j < Sd::CH; ++j) atomicAdd(&dbl[cg * Sd::CH + j], dbs[j]);
        __syncthreads();
        if (threadIdx.x < CBM) atomicAdd(p.db + c0 + threadIdx.x, dbl[threadIdx.x]);
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
    hipLaunchKernelGGL((dwconv7_mfma_kernel<TX, TY, FLIP>), dim3(grid), dim3(NTHR), lds, st, p);
    return 0;
}

template <typename TX, typename TDY> int launch_wgrad(const MwP& p, int grid, hipStream_t st) {
    const int lds = WX_BYTES + WD_BYTES + CBM * (int)sizeof(float);
    static bool ready = false;
    if (!ready) {
        if (int rc = set_lds(dwconv7_mfma_wgrad_kernel<TX, TDY>, lds)) return rc;
        ready = true;
    }
    hipLaunchKernelGGL((dwconv7_mfma_wgrad_kernel<TX, TDY>), dim3(grid), dim3(NTHR), lds, st, p);
    return 0;
}

int cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
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
    p.tiles_h = cdiv(a->H, MT); p.tiles_w = cdiv(a->W, MT);
    const int64_t ntile = (int64_t)a->B * p.tiles_h * p.tiles_w;
    LNX_CHECK(ntile < (1ll << 31), "lnx_dwconv7_fwd: too many tiles");
    p.ntile = (int)ntile;
    // one resident workgroup per CU (register-resident Toeplitz operands): each walks a contiguous run of tiles
    const int cblocks = a->C / CBM;
    int chunks = cus() / cblocks;
    if (chunks < 1) chunks = 1;
    if (chunks > p.ntile) chunks = p.ntile;
    p.tiles_per_wg = cdiv(p.ntile, chunks);
    p.chunks = cdiv(p.ntile, p.tiles_per_wg);
    const int grid = p.chunks * cblocks;
    LNX_CHECK(!(a->res && a->y_dtype != LNX_F32), "lnx_dwconv7_fwd: a residual needs an fp32 output");
    const int code = a->x_dtype * 2 + a->y_dtype + (a->flip ? 4 : 0);
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
    p.tiles_h = cdiv(a->H, WH); p.tiles_w = cdiv(a->W, WW); p.cblocks = a->C / CBM;
    const int64_t ntile = (int64_t)a->B * p.tiles_h * p.tiles_w;
    LNX_CHECK(ntile < (1ll << 31), "lnx_dwconv7_wgrad: too many tiles");
    p.ntile = (int)ntile;
    int walkers = cus() / p.cblocks;
    if (walkers < 1) walkers = 1;
    if (walkers > p.ntile) walkers = p.ntile;
    walkers = cdiv(p.ntile, cdiv(p.ntile, walkers));  // same longest walk, no idle walkers
    const int grid = walkers * p.cblocks;
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
