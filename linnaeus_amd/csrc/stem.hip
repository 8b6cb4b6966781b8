// The stem in one kernel (bf16 compute type): 4x4 / stride-4 patchify convolution + bias + channels-first LayerNorm.
// Reference: nn.Sequential(nn.Conv2d(in_chans, dims[0], 4, 4), LayerNorm(dims[0], eps=1e-6, "channels_first")),
// mFormerV1.py:145-148, under autocast (bf16 operands, the convolution's output rounded to bf16, fp32 statistics).
//
// It replaced im2col (read the image, write the patch matrix) -> GEMM with K = 48 of 64 (read the patches, write the bf16
// pre-norm tensor; one K step, so the register-staged 128x128 kernel: 143 us at sm / B = 256) -> LayerNorm (read it, write
// the fp32 stream): 293 us and 0.98 GB of traffic for 7.4 GFLOP.  Here a wave owns 16 output pixels at a time:
//   * lane (s, g) gathers pixel s's taps k = 8g .. 8g+7 (two float4 of the NCHW image: k = c*16 + kh*4 + kw, so 4 consecutive
//     k are 4 consecutive floats of one image row) and k = 32 + 8g .. (channel 2, only g < 2 when in_chans = 3): these ARE
//     the MFMA operand of v_mfma_f32_16x16x32_bf16 -- no LDS -- and, stored 16 bytes a lane, the patch matrix the weight
//     gradient of the backward reads;
//   * the weight fragments stay in registers for the whole kernel; lane (s, g) ends with channels 16 ni + 4 g .. + 3 (ni < C / 16) of
//     pixel s, so the LayerNorm statistics are a sum over the lane's values and the four g-lanes of the pixel (two shuffles), the
//     fp32 row is written 64 contiguous bytes per pixel and store instruction, and the bf16 pre-norm row in 16-byte pieces after a
//     swap between neighbouring g-lanes;
//   * the next tile's pixels are fetched before this tile's stores are issued (one in-order memory counter: fetched after them, the
//     loads would wait for the stores' acknowledgements), and NO memory operation of the tile loop sits under a branch, so that the
//     compiler can count that counter instead of draining it: 265 -> 160 us at sm / B = 256 (4.5 TB/s).
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;

__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) {
    Vec16<bf16_t> v;
    v.set(0, a.x); v.set(1, a.y); v.set(2, a.z); v.set(3, a.w);
    v.set(4, b.x); v.set(5, b.y); v.set(6, b.z); v.set(7, b.w);
    return v.raw;
}

__device__ __forceinline__ void lds_read16(f32x4_t& dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }

struct StemP {
    const float* x;
    const bf16_t* w;  // [Cout][64], k = c*16 + kh*4 + kw
    const float *bias, *lnw, *lnb;
    bf16_t* patches;  // [M, 64] or nullptr
    bf16_t* pre;      // [M, Cout] or nullptr
    float* y;         // [M, Cout]
    float *mean, *rstd;
    int B, Cin, H, W, Ho, Wo, M;
    float eps;
};

// TRAIN: also write the patch matrix, the bf16 pre-norm rows and the row statistics (all four, or none: every memory operation of the
// tile loop is then unconditional, which is what lets the compiler count the in-order memory counter instead of draining it)
template <int NT, bool TRAIN>  // Cout = 16 NT
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT <= 6 ? 3 : (NT <= 8 ? 2 : 1)))) void stem_fwd_kernel(const StemP p) {
    constexpr int Cout = 16 * NT;
    __shared__ __attribute__((aligned(16))) float prm[3][Cout];  // conv bias | LayerNorm weight | LayerNorm bias
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < Cout; i += 256) {
        prm[0][i] = p.bias[i];
        prm[1][i] = p.lnw[i];
        prm[2][i] = p.lnb[i];
    }
    // weight fragments: MFMA row j of tile ni is output channel 16 ni + j, so lane (s, g) ends with channels 16 ni + 4 g .. + 3 of
    // pixel s -- the four g-lanes of a pixel write 64 contiguous bytes of the fp32 row per store instruction
    uint4 wf[NT][2];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const bf16_t* wr = p.w + (int64_t)(16 * ni + s) * 64 + 8 * g;
        const uint4 lo = ld16(wr), hi = ld16(wr + 32);  // columns beyond in_chans * 16 are padding: whatever they hold must not count
        wf[ni][0] = 8 * g < p.Cin * 16 ? lo : make_uint4(0u, 0u, 0u, 0u);
        wf[ni][1] = 32 + 8 * g < p.Cin * 16 ? hi : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // the three parameter rows are read from LDS per tile (inline asm: as plain loads the compiler hoists all 3 NT of them out of the
    // tile loop -- 72 registers at C = 96, half the occupancy)
    const uint32_t prm_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&prm[0][0] + (uint32_t)(4 * g * 4);
    const int64_t plane = (int64_t)p.H * p.W;
    // grid-stride over 16-pixel tiles: the launch is one round of resident workgroups (no partly filled last round), and the waves
    // of the chip work on neighbouring tiles at any moment
    const int tile0 = blockIdx.x * 4 + wave, tstride = gridDim.x * 4;
    const int ntiles = (p.M + 15) / 16;
    const int nt = tile0 < ntiles ? (ntiles - tile0 + tstride - 1) / tstride : 0;  // wave-uniform
    // taps (c, kh) = idx / 4, idx % 4 for idx = 2g, 2g + 1 (k step 0) and 8 + 2g, 9 + 2g (k step 1); unconditional loads, select after
    auto fetch = [&](int t, float4 (&v)[4]) __attribute__((always_inline)) {
        const int m = min((tile0 + t * tstride) * 16 + s, p.M - 1);
        const int wo = m % p.Wo, t2 = m / p.Wo;
        const int ho = t2 % p.Ho, b = t2 / p.Ho;
        const float* px = p.x + (int64_t)b * p.Cin * plane + (int64_t)(4 * ho) * p.W + 4 * wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = (j >> 1) * 8 + 2 * g + (j & 1);
            const bool in = idx < p.Cin * 4;
            const int ic = in ? idx : 0;
            const float4 q = *reinterpret_cast<const float4*>(px + (int64_t)(ic >> 2) * plane + (int64_t)(ic & 3) * p.W);
            const uint32_t keep = in ? 0xffffffffu : 0u;  // a mask, not a select: a select lets the compiler move the load under a branch
            v[j] = make_float4(__uint_as_float(__float_as_uint(q.x) & keep), __uint_as_float(__float_as_uint(q.y) & keep),
                               __uint_as_float(__float_as_uint(q.z) & keep), __uint_as_float(__float_as_uint(q.w) & keep));
        }
    };
    float4 v[4];
    if (nt > 0) fetch(0, v);
    for (int t = 0; t < nt; ++t) {
        const int m0 = (tile0 + t * tstride) * 16;
        // pixels beyond M (last tile only) are clamped to M - 1 in fetch(): such a lane computes pixel M - 1 again and stores the same
        // bytes to the same row as its live twin -- no predicate anywhere in the loop
        const int m = min(m0 + s, p.M - 1);
        const uint4 a0 = pack8(v[0], v[1]), a1 = pack8(v[2], v[3]);
        // the next tile's pixels travel while this one is multiplied, normalised and stored: fetched after the stores below they
        // would wait for those stores' acknowledgements first (one in-order counter), a second memory round trip per tile
        fetch(min(t + 1, nt - 1), v);
        if constexpr (TRAIN) {
            st16(p.patches + (int64_t)m * 64 + 8 * g, a0);
            st16(p.patches + (int64_t)m * 64 + 32 + 8 * g, a1);
        }
        f32x4_t acc[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, wf[ni][0]), __builtin_bit_cast(bf16x8v, a0), acc[ni], 0, 0, 0);
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, wf[ni][1]), __builtin_bit_cast(bf16x8v, a1), acc[ni], 0, 0, 0);
        }
        // lane (s, g): channels 16 ni + 4 g + r of pixel s.  The convolution's output is a bf16 tensor (autocast); the norm reads that.
        float val[NT][4];
        float sum = 0.f;
        f32x4_t cbq[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) lds_read16(cbq[ni], prm_addr + (uint32_t)((16 * ni) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // nothing that uses the values may be scheduled above the wait
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const float cbv[4] = {cbq[ni][0], cbq[ni][1], cbq[ni][2], cbq[ni][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float q = (float)(bf16_t)(acc[ni][r] + cbv[r]);
                val[ni][r] = q;
                sum += q;
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mu = sum / (float)Cout;
        float sq = 0.f;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) sq = fmaf(val[ni][r] - mu, val[ni][r] - mu, sq);
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rs = rsqrtf(sq / (float)Cout + p.eps);
        if constexpr (TRAIN) {
            // bf16 row: lanes g and g ^ 1 swap halves so that the even one writes channels 16 ni + 4 g .. + 7 of the even tiles and
            // the odd one those of the odd tiles -- one 16-byte store per lane and tile pair (NT is even), address and data by select
            bf16_t* po = p.pre + (int64_t)m * Cout + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);
            const bool even = (g & 1) == 0;
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) {
                uint2 mine[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16_t* hh = reinterpret_cast<bf16_t*>(&mine[h]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) hh[r] = (bf16_t)val[2 * q + h][r];
                }
                const uint2 give = even ? mine[1] : mine[0], keep = even ? mine[0] : mine[1];
                uint2 got;
                got.x = (uint32_t)__shfl_xor((int)give.x, 16, 64);
                got.y = (uint32_t)__shfl_xor((int)give.y, 16, 64);
                const uint2 lo = even ? keep : got, hi = even ? got : keep;
                st16(po + 32 * q, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
        }
        {
            float* yo = p.y + (int64_t)m * Cout + 4 * g;
#pragma unroll
            for (int ni = 0; ni < NT; ni += 2) {
                f32x4_t lw[2], lb[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    lds_read16(lw[h], prm_addr + (uint32_t)((Cout + 16 * (ni + h)) * 4));
                    lds_read16(lb[h], prm_addr + (uint32_t)((2 * Cout + 16 * (ni + h)) * 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<float4*>(yo + 16 * (ni + h)) =
                        make_float4((val[ni + h][0] - mu) * rs * lw[h][0] + lb[h][0], (val[ni + h][1] - mu) * rs * lw[h][1] + lb[h][1],
                                    (val[ni + h][2] - mu) * rs * lw[h][2] + lb[h][2], (val[ni + h][3] - mu) * rs * lw[h][3] + lb[h][3]);
            }
            if constexpr (TRAIN) {
                if (g == 0) {
                    p.mean[m] = mu;
                    p.rstd[m] = rs;
                }
            }
        }
    }
}

}  // namespace

extern "C" int lnx_stem_fwd_ok(int dtype, int Cin, int H, int W, int Cout) {
    static const bool off = getenv("LNX_NO_FUSED_STEM") != nullptr;
    return !off && dtype == LNX_BF16 && Cin >= 1 && Cin <= 4 && H % 4 == 0 && W % 4 == 0 && (Cout == 96 || Cout == 128 || Cout == 192 || Cout == 256);
}

extern "C" int lnx_stem_fwd(const lnx_stem_args* a, void* stream) {
    LNX_CHECK(a && a->x && a->w && a->bias && a->ln_w && a->ln_b && a->y, "lnx_stem_fwd: null operand");
    LNX_CHECK(lnx_stem_fwd_ok(LNX_BF16, a->Cin, a->H, a->W, a->Cout) || getenv("LNX_NO_FUSED_STEM"), "lnx_stem_fwd: unsupported geometry Cin=%d %dx%d Cout=%d", a->Cin, a->H,
              a->W, a->Cout);
    LNX_CHECK((a->mean == nullptr) == (a->pre == nullptr) && (a->rstd == nullptr) == (a->pre == nullptr) && (a->patches == nullptr) == (a->pre == nullptr),
              "lnx_stem_fwd: patches, pre, mean and rstd are written together (a training plan) or not at all");
    LNX_CHECK((((uintptr_t)a->x) & 15) == 0 && (((uintptr_t)a->w) & 15) == 0 && (((uintptr_t)a->y) & 15) == 0 && (((uintptr_t)a->patches) & 15) == 0 &&
                  (((uintptr_t)a->pre) & 15) == 0,
              "lnx_stem_fwd: operands must be 16-byte aligned");
    StemP p;
    p.x = a->x; p.w = (const bf16_t*)a->w; p.bias = a->bias; p.lnw = a->ln_w; p.lnb = a->ln_b;
    p.patches = (bf16_t*)a->patches; p.pre = (bf16_t*)a->pre; p.y = a->y; p.mean = a->mean; p.rstd = a->rstd;
    p.B = a->B; p.Cin = a->Cin; p.H = a->H; p.W = a->W; p.Ho = a->H / 4; p.Wo = a->W / 4;
    p.M = a->B * p.Ho * p.Wo;
    p.eps = a->eps;
    const int tiles = cdiv(p.M, 16);
    int grid = cdiv(tiles, 4);
    const int resident = 256 * (a->Cout == 96 ? 3 : (a->Cout == 128 ? 2 : 1));  // workgroups the chip holds at this kernel's register count
    if (grid > resident) grid = resident;
    hipStream_t st = (hipStream_t)stream;
#define STEM_(NTV)                                                                                         \
    do {                                                                                                  \
        if (a->pre) hipLaunchKernelGGL((stem_fwd_kernel<NTV, true>), dim3(grid), dim3(256), 0, st, p);    \
        else hipLaunchKernelGGL((stem_fwd_kernel<NTV, false>), dim3(grid), dim3(256), 0, st, p);          \
    } while (0)
    switch (a->Cout) {
        case 96: STEM_(6); break;
        case 128: STEM_(8); break;
        case 192: STEM_(12); break;
        default: STEM_(16); break;
    }
#undef STEM_
    LNX_LAUNCH_CHECK();
    return 0;
}
