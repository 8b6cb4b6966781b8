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
//   * the weight fragments stay in registers for the whole kernel, with the output channels PERMUTED over the MFMA rows so
//     that lane (s, g) ends up with the 4 NT consecutive channels g 4NT .. of pixel s: the LayerNorm statistics are a sum over
//     the lane's values and the four g-lanes of the pixel (two shuffles), and every store is 16 contiguous bytes.
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

struct StemP {
    const float* x;
    const bf16_t* w;  // [Cout][64], k = c*16 + kh*4 + kw
    const float *bias, *lnw, *lnb;
    bf16_t* patches;  // [M, 64] or nullptr
    bf16_t* pre;      // [M, Cout] or nullptr
    float* y;         // [M, Cout]
    float *mean, *rstd;
    int B, Cin, H, W, Ho, Wo, M, tiles_per_wave;
    float eps;
};

template <int NT>  // Cout = 16 NT
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemP p) {
    constexpr int CPL = 4 * NT;  // channels per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const int Cout = 16 * NT;
    // weight fragments: MFMA row j of tile ni is output channel (j >> 2) * CPL + 4 ni + (j & 3)
    uint4 wf[NT][2];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const bf16_t* wr = p.w + (int64_t)((s >> 2) * CPL + 4 * ni + (s & 3)) * 64 + 8 * g;
        const uint4 lo = ld16(wr), hi = ld16(wr + 32);  // columns beyond in_chans * 16 are padding: whatever they hold must not count
        wf[ni][0] = 8 * g < p.Cin * 16 ? lo : make_uint4(0u, 0u, 0u, 0u);
        wf[ni][1] = 32 + 8 * g < p.Cin * 16 ? hi : make_uint4(0u, 0u, 0u, 0u);
    }
    float cb[CPL], lw[CPL], lb[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        cb[i] = p.bias[g * CPL + i];
        lw[i] = p.lnw[g * CPL + i];
        lb[i] = p.lnb[g * CPL + i];
    }
    const int64_t plane = (int64_t)p.H * p.W;
    const int tile0 = (blockIdx.x * 4 + wave) * p.tiles_per_wave;
    for (int t = 0; t < p.tiles_per_wave; ++t) {
        const int m0 = (tile0 + t) * 16;
        if (m0 >= p.M) break;  // wave-uniform
        const int m = min(m0 + s, p.M - 1);
        const int wo = m % p.Wo, t2 = m / p.Wo;
        const int ho = t2 % p.Ho, b = t2 / p.Ho;
        // taps (c, kh) = idx / 4, idx % 4 for idx = 2g, 2g + 1 (k step 0) and 8 + 2g, 9 + 2g (k step 1)
        const float* px = p.x + (int64_t)b * p.Cin * plane + (int64_t)(4 * ho) * p.W + 4 * wo;
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = (j >> 1) * 8 + 2 * g + (j & 1);
            const bool in = idx < p.Cin * 4;
            const int ic = in ? idx : 0;  // unconditional load, select after
            const float4 q = *reinterpret_cast<const float4*>(px + (int64_t)(ic >> 2) * plane + (int64_t)(ic & 3) * p.W);
            v[j] = in ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const uint4 a0 = pack8(v[0], v[1]), a1 = pack8(v[2], v[3]);
        const bool live = m0 + s < p.M;
        if (p.patches && live) {
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
        // lane (s, g): channels g CPL + 4 ni + r of pixel s.  The convolution's output is a bf16 tensor (autocast); the norm
        // reads that.
        float val[CPL];
        float sum = 0.f;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float q = (float)(bf16_t)(acc[ni][r] + cb[4 * ni + r]);
                val[4 * ni + r] = q;
                sum += q;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mu = sum / (float)Cout;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) sq = fmaf(val[i] - mu, val[i] - mu, sq);
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rs = rsqrtf(sq / (float)Cout + p.eps);
        if (!live) continue;
        if (p.pre) {
            bf16_t* po = p.pre + (int64_t)m * Cout + g * CPL;
#pragma unroll
            for (int i = 0; i < CPL; i += 8) {
                if (i + 8 <= CPL) {
                    Vec16<bf16_t> o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.set(j, val[i + j]);
                    st16(po + i, o.raw);
                } else {  // CPL = 24: a last group of 4 (8 bytes)
                    uint2 o;
                    bf16_t* h = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = (bf16_t)val[i + j];
                    *reinterpret_cast<uint2*>(po + i) = o;
                }
            }
        }
        float* yo = p.y + (int64_t)m * Cout + g * CPL;
#pragma unroll
        for (int i = 0; i < CPL; i += 4)
            *reinterpret_cast<float4*>(yo + i) = make_float4((val[i] - mu) * rs * lw[i] + lb[i], (val[i + 1] - mu) * rs * lw[i + 1] + lb[i + 1],
                                                             (val[i + 2] - mu) * rs * lw[i + 2] + lb[i + 2], (val[i + 3] - mu) * rs * lw[i + 3] + lb[i + 3]);
        if (g == 0 && p.mean) {
            p.mean[m] = mu;
            p.rstd[m] = rs;
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
    LNX_CHECK((a->mean == nullptr) == (a->rstd == nullptr), "lnx_stem_fwd: mean and rstd go together");
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
    p.tiles_per_wave = tiles >= 8192 ? 8 : 1;  // amortise the weight-fragment loads when there is work for every CU anyway
    const int grid = cdiv(tiles, 4 * p.tiles_per_wave);
    hipStream_t st = (hipStream_t)stream;
    switch (a->Cout) {
        case 96: hipLaunchKernelGGL((stem_fwd_kernel<6>), dim3(grid), dim3(256), 0, st, p); break;
        case 128: hipLaunchKernelGGL((stem_fwd_kernel<8>), dim3(grid), dim3(256), 0, st, p); break;
        case 192: hipLaunchKernelGGL((stem_fwd_kernel<12>), dim3(grid), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((stem_fwd_kernel<16>), dim3(grid), dim3(256), 0, st, p); break;
    }
    LNX_LAUNCH_CHECK();
    return 0;
}
