// Depthwise 7x7 convolution (padding 3, NHWC) for gfx950: forward / data-gradient and
// weight-gradient kernels.  Reference: nn.Conv2d(dim, dim, 7, padding=3, groups=dim),
// blocks/convnext.py:56-58.
//
// HBM-bound op: algorithmic traffic is one read of x and one write of y per element
// (49 MACs per element).  A workgroup walks a sequence of 8x16-pixel x 32-channel tiles; the
// tile's input with its 3-pixel halo is staged in LDS (channel-innermost: a wave reads 32
// consecutive channels = conflict-free ds_read_b32) and each thread produces a 16-pixel output
// row segment from registers (22 + 7 LDS reads per 112 FMAs, ky loop kept rolled so only one
// input row is live).  The NEXT tile's halo is fetched global->registers before the current
// tile's FMAs and written to LDS after them, so HBM/L2 latency hides under the arithmetic
// (issue-early / write-late staging) instead of serialising load -> compute -> store.
//
// Tile: 8 rows x 16 cols x 32 channels of output per 256-thread workgroup
// (thread = (channel lane 0..31, row 0..7)); LDS 14 x 22 x 32 fp32 = 38.5 KiB + 6 KiB taps.
#include "common.hpp"
#include "../../include/lnx.h"
#include "dwconv_mfma.hpp"

namespace {

constexpr int TH = 8, TW = 16, CB = 32;
constexpr int IH = TH + 6, IW = TW + 6;
constexpr int NV = IH * IW * (CB / 4);         // float4 slots of a halo tile
constexpr int NPRE = (NV + 255) / 256;         // per-thread prefetch registers
constexpr int NDY = TH * TW * (CB / 4) / 256;  // float4 slots per thread of an output-shaped tile

struct DwP {
    const void* x;
    const float* w49;
    const float* bias;
    const float* res;
    void* y;
    int B, H, W, C;
    int flip;
    int tiles_h, tiles_w, cblocks;
    int tiles_per_wg;  // consecutive tiles (same channel block) one workgroup walks
};

template <typename TX> struct Raw4;  // 4 consecutive channels as loaded
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };

__device__ __forceinline__ float4 widen(const float4& v) { return v; }
__device__ __forceinline__ float4 widen(const uint2& r) {
    const bf16_t* h = reinterpret_cast<const bf16_t*>(&r);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void zero_raw(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero_raw(uint2& v) { v = make_uint2(0u, 0u); }

// tile index -> (b, th, tw); tiles of one channel block are numbered ((b*tiles_h + th)*tiles_w + tw)
struct TilePos {
    int b, h0, w0;
};
__device__ __forceinline__ TilePos tile_pos(int t, int tiles_h, int tiles_w) {
    TilePos q;
    const int tw = t % tiles_w;
    const int t2 = t / tiles_w;
    q.b = t2 / tiles_h;
    q.h0 = (t2 % tiles_h) * TH;
    q.w0 = tw * TW;
    return q;
}

template <typename TX>
__device__ __forceinline__ void halo_issue(typename Raw4<TX>::type (&pre)[NPRE], const TX* __restrict__ x, const TilePos& q, int c0, int H, int W, int C) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int i = threadIdx.x + 256 * k;
        zero_raw(pre[k]);
        if (i < NV) {
            const int cv = i % (CB / 4);
            const int pix = i / (CB / 4);
            const int iw = pix % IW, ih = pix / IW;
            const int h = q.h0 + ih - 3, w = q.w0 + iw - 3;
            if (h >= 0 && h < H && w >= 0 && w < W)
                pre[k] = *reinterpret_cast<const typename Raw4<TX>::type*>(x + (((int64_t)q.b * H + h) * W + w) * C + c0 + 4 * cv);
        }
    }
}

template <typename TX>
__device__ __forceinline__ void halo_commit(float* __restrict__ tile, const typename Raw4<TX>::type (&pre)[NPRE]) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NV) *reinterpret_cast<float4*>(tile + 4 * i) = widen(pre[k]);  // tile[pix*CB + 4*cv] == tile[4*i]
    }
}

// Diagnostic build only (-DDW_STAMP, tools/build_stamp.sh): per-phase s_memtime sums of wave 0 of workgroup 0
#ifdef DW_STAMP
__device__ unsigned long long g_dw_stamp[8];
#define DW_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define DW_T(i) do { } while (0)
#endif

template <typename TX, typename TY, bool FLIP>
__global__ __launch_bounds__(256) void dwconv7_kernel(const DwP p) {
    __shared__ __attribute__((aligned(16))) float tile[IH * IW * CB];
    __shared__ float wtile[49 * CB];
    const int ntile = p.B * p.tiles_h * p.tiles_w;
    const int chunks = (ntile + p.tiles_per_wg - 1) / p.tiles_per_wg;
    const int cb = blockIdx.x / chunks;
    const int t_begin = (blockIdx.x % chunks) * p.tiles_per_wg;
    const int t_end = min(ntile, t_begin + p.tiles_per_wg);
    const int c0 = cb * CB;
    const int cl = threadIdx.x & 31;
    const int r = threadIdx.x >> 5;
    const int c = c0 + cl;
    const TX* xg = reinterpret_cast<const TX*>(p.x);

    for (int i = threadIdx.x; i < 49 * CB; i += 256) {
        const int t = i / CB, cc = i % CB;
        wtile[i] = p.w49[(FLIP ? 48 - t : t) * p.C + c0 + cc];
    }
    const float bv = p.bias ? p.bias[c] : 0.f;

    typename Raw4<TX>::type pre[NPRE];
    halo_issue<TX>(pre, xg, tile_pos(t_begin, p.tiles_h, p.tiles_w), c0, p.H, p.W, p.C);
    halo_commit<TX>(tile, pre);
    __syncthreads();

#ifdef DW_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    for (int t = t_begin; t < t_end; ++t) {
        const TilePos q = tile_pos(t, p.tiles_h, p.tiles_w);
        const bool more = t + 1 < t_end;
        if (more) halo_issue<TX>(pre, xg, tile_pos(t + 1, p.tiles_h, p.tiles_w), c0, p.H, p.W, p.C);
        DW_T(0);

        float acc[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[j] = bv;
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
            float wv[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) wv[kx] = wtile[(ky * 7 + kx) * CB + cl];
            float in[IW];
            const float* row = tile + ((r + ky) * IW) * CB + cl;
#pragma unroll
            for (int j = 0; j < IW; ++j) in[j] = row[j * CB];
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) acc[j] = fmaf(wv[kx], in[j + kx], acc[j]);
        }
        DW_T(1);
        const int h = q.h0 + r;
        if (h < p.H) {
            const int64_t off0 = (((int64_t)q.b * p.H + h) * p.W + q.w0) * p.C + c;
            if (p.res) {
                // every residual load before the first store: the compiler must assume y aliases res, so an
                // interleaved load/store sequence would cost one memory round trip per pixel
                float rv[TW];
#pragma unroll
                for (int j = 0; j < TW; ++j) rv[j] = (q.w0 + j < p.W) ? p.res[off0 + (int64_t)j * p.C] : 0.f;
#pragma unroll
                for (int j = 0; j < TW; ++j) acc[j] += rv[j];
            }
#pragma unroll
            for (int j = 0; j < TW; ++j)
                if (q.w0 + j < p.W) reinterpret_cast<TY*>(p.y)[off0 + (int64_t)j * p.C] = from_f<TY>(acc[j]);
        }
        DW_T(2);
        __syncthreads();  // every wave is done reading this tile
        DW_T(3);
        if (more) {
            halo_commit<TX>(tile, pre);
            DW_T(4);
            __syncthreads();
            DW_T(5);
        }
    }
#ifdef DW_STAMP
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_dw_stamp[i] = tsum[i];
#endif
}

#ifdef DW_STAMP
extern "C" int lnx_dbg_dwconv_stamps(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dw_stamp), 64); }
#endif

struct DwWgP {
    const void* x;
    const void* dy;
    float* dw;
    float* db;
    int B, H, W, C;
    int tiles_h, tiles_w, cblocks, ntile;  // ntile = B * tiles_h * tiles_w
};

template <typename TDY>
__device__ __forceinline__ void dy_issue(typename Raw4<TDY>::type (&pre)[NDY], const TDY* __restrict__ dy, const TilePos& q, int c0, int H, int W, int C) {
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int cv = i % (CB / 4);
        const int pix = i / (CB / 4);
        const int j = pix % TW, rr = pix / TW;
        const int h = q.h0 + rr, w = q.w0 + j;
        zero_raw(pre[k]);
        if (h < H && w < W) pre[k] = *reinterpret_cast<const typename Raw4<TDY>::type*>(dy + (((int64_t)q.b * H + h) * W + w) * C + c0 + 4 * cv);
    }
}

// dW[c, ky, kx] += sum dY[b,h,w,c] * x[b,h+ky-3,w+kx-3,c];   db[c] += sum dY
// Each workgroup owns one 32-channel block and walks tiles grid-stride with the same
// register-prefetch pipeline, keeping its 49 tap sums per thread in registers; one LDS
// reduction + one atomic per tap at the end.
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
    const TX* xg = reinterpret_cast<const TX*>(p.x);
    const TDY* dg = reinterpret_cast<const TDY*>(p.dy);

    float acc[49];
#pragma unroll
    for (int t = 0; t < 49; ++t) acc[t] = 0.f;
    float accb = 0.f;

    typename Raw4<TX>::type pre[NPRE];
    typename Raw4<TDY>::type pdy[NDY];
    if (walker < p.ntile) {
        const TilePos q0 = tile_pos(walker, p.tiles_h, p.tiles_w);
        halo_issue<TX>(pre, xg, q0, c0, p.H, p.W, p.C);
        dy_issue<TDY>(pdy, dg, q0, c0, p.H, p.W, p.C);
        halo_commit<TX>(tile, pre);
#pragma unroll
        for (int k = 0; k < NDY; ++k) *reinterpret_cast<float4*>(dyt + 4 * (threadIdx.x + 256 * k)) = widen(pdy[k]);
    }
    __syncthreads();
    for (int t = walker; t < p.ntile; t += nwalk) {
        const bool more = t + nwalk < p.ntile;
        if (more) {
            const TilePos qn = tile_pos(t + nwalk, p.tiles_h, p.tiles_w);
            halo_issue<TX>(pre, xg, qn, c0, p.H, p.W, p.C);
            dy_issue<TDY>(pdy, dg, qn, c0, p.H, p.W, p.C);
        }
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
        __syncthreads();
        if (more) {
            halo_commit<TX>(tile, pre);
#pragma unroll
            for (int k = 0; k < NDY; ++k) *reinterpret_cast<float4*>(dyt + 4 * (threadIdx.x + 256 * k)) = widen(pdy[k]);
            __syncthreads();
        }
    }
    // reduce over the 8 row-threads of each channel through LDS (reusing `tile`), two passes of 25
    // slots: taps 0..24, then taps 25..48 + the bias sum
    float* red = tile;
#pragma unroll
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
    // bf16 compute: the matrix-core kernels (dwconv_mfma.hip); fp32 (strict-parity mode) stays on the VALU kernels below
    if (lnx_dwconv_mfma_enabled() && (a->x_dtype == LNX_BF16 || a->y_dtype == LNX_BF16) && !(a->res && a->y_dtype != LNX_F32))
        return lnx_dwconv7_mfma_fwd(a, (hipStream_t)stream);
    DwP p;
    p.x = a->x; p.w49 = a->w49; p.bias = a->bias; p.res = a->res; p.y = a->y;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C; p.flip = a->flip;
    p.tiles_h = cdiv(a->H, TH); p.tiles_w = cdiv(a->W, TW); p.cblocks = a->C / CB;
    // each workgroup walks the tiles of (up to) one image; with fewer than ~3 workgroups per CU the
    // walk is shortened so the chip stays full
    const int64_t ntile = (int64_t)a->B * p.tiles_h * p.tiles_w;
    int per = p.tiles_h * p.tiles_w;
    while (per > 1 && (int64_t)cdiv(ntile, per) * p.cblocks < 3 * 256) per = (per + 1) / 2;
    p.tiles_per_wg = per;
    const int64_t grid = (int64_t)cdiv(ntile, per) * p.cblocks;
    LNX_CHECK(grid < (1ll << 31), "lnx_dwconv7_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const int code = a->x_dtype * 2 + a->y_dtype + (a->flip ? 4 : 0);
#define DWL(TX, TY, F) hipLaunchKernelGGL((dwconv7_kernel<TX, TY, F>), dim3((int)grid), dim3(256), 0, st, p)
    switch (code) {
        case 0: DWL(float, float, false); break;
        case 1: DWL(float, bf16_t, false); break;
        case 2: DWL(bf16_t, float, false); break;
        case 3: DWL(bf16_t, bf16_t, false); break;
        case 4: DWL(float, float, true); break;
        case 5: DWL(float, bf16_t, true); break;
        case 6: DWL(bf16_t, float, true); break;
        case 7: DWL(bf16_t, bf16_t, true); break;
        default: LNX_CHECK(false, "lnx_dwconv7_fwd: bad dtypes");
    }
#undef DWL
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_dwconv7_wgrad(const lnx_dwconv_wgrad_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->dy && a->dw, "lnx_dwconv7_wgrad: null operand");
    LNX_CHECK(a->B > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->C % CB == 0, "lnx_dwconv7_wgrad: bad shape");
    if (lnx_dwconv_mfma_enabled() && (a->x_dtype == LNX_BF16 || a->dy_dtype == LNX_BF16)) return lnx_dwconv7_mfma_wgrad(a, (hipStream_t)stream);
    DwWgP p;
    p.x = a->x; p.dy = a->dy; p.dw = a->dw; p.db = a->db;
    p.B = a->B; p.H = a->H; p.W = a->W; p.C = a->C;
    p.tiles_h = cdiv(a->H, TH); p.tiles_w = cdiv(a->W, TW); p.cblocks = a->C / CB;
    p.ntile = a->B * p.tiles_h * p.tiles_w;
    int walkers = 768 / p.cblocks;
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
