// C[M <= 256, N] = A[M, K] . W[N, K]^T (+ bias, + fp32 residual) for the M = batch products of the model's tail: the
// classification heads (reference: ClassificationHead / ConditionalClassifier fc, mFormerV1.py:536-541) and their data
// gradients.  With 256 rows the tiled kernels launch 2 x ceil(N / 128) workgroups whose K loops (stage, barrier, multiply,
// twelve times for K = 384) are pure latency: 21-28 us per launch for 0.2 GFLOP, eight launches in the middle of every step
// with nothing to overlap them.  Here a wave owns a 32 x 32 output tile and reads its fragments straight from global memory
// (both operands are small and L2-resident; lane (s, g) takes 16 bytes = 8 k of one row, which is the MFMA operand as it
// stands), a whole batch of 4 k-steps in flight before the first MFMA: no LDS, no barrier, M / 32 x N / 32 independent waves.
#include "gemm_common.hpp"

namespace lnxg {

namespace {
constexpr int SK_KU = 4;  // k-steps of 32 per batch of loads: 16 loads of 16 bytes per lane in flight

template <bool OUT_F32>
__global__ __launch_bounds__(64) void gemm_nt_skinny_kernel(const GemmP p) {
    typedef bf16_t T;
    const int lane = threadIdx.x, s = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* W = reinterpret_cast<const T*>(p.W);
    // clamped rows: what they produce is not stored
    const T* ar[2] = {A + (int64_t)min(m0 + s, p.M - 1) * p.lda + 8 * g, A + (int64_t)min(m0 + 16 + s, p.M - 1) * p.lda + 8 * g};
    const T* wr[2] = {W + (int64_t)min(n0 + s, p.N - 1) * p.ldw + 8 * g, W + (int64_t)min(n0 + 16 + s, p.N - 1) * p.ldw + 8 * g};
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nks = (p.K + 31) / 32;
    for (int k0 = 0; k0 < nks; k0 += SK_KU) {
        uint4 af[SK_KU][2], wf[SK_KU][2];
#pragma unroll
        for (int u = 0; u < SK_KU; ++u) {
            // K is a multiple of 8: a lane's chunk lies wholly inside or wholly outside the row.  Outside: read chunk 0
            // (unconditional load) and zero the W fragment, which zeroes the product.
            const int k = (k0 + u) * 32;
            const bool in = k + 8 * g < p.K;
            const int ko = in ? k : -8 * g;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[u][i] = ld16(ar[i] + ko);
                wf[u][i] = ld16(wr[i] + ko);
            }
        }
#pragma unroll
        for (int u = 0; u < SK_KU; ++u) {
            const bool in = (k0 + u) * 32 + 8 * g < p.K;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const uint4 w = in ? wf[u][ni] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mfma<T>::run(acc[ni][mi], w, af[u][mi]);  // D[n = 4g + r][m = s]
            }
        }
    }
    // lane (s, g): row m = m0 + 16 mi + s, columns n0 + 16 ni + 4 g .. + 3
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 16 * mi + s;
        if (m >= p.M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + 16 * ni + 4 * g;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r] + (p.bias && n + r < p.N ? p.bias[n + r] : 0.f);
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] += p.res[(int64_t)m * p.ldres + n + r];
            }
            if constexpr (OUT_F32) {
                float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
                if (n + 3 < p.N && (p.ldc & 3) == 0) {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) c[r] = v[r];
                }
            } else {
                T* c = reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) c[r] = (T)v[r];
            }
        }
    }
}
}  // namespace

bool nt_skinny_ok(const GemmP& p, int dtype, bool out_f32) {
    static const bool off = getenv("LNX_NT_SKINNY") != nullptr && atoi(getenv("LNX_NT_SKINNY")) == 0;
    if (off || dtype != LNX_BF16 || p.M > 256) return false;
    if (p.a_mode != LNX_ADDR_PLAIN || p.c_mode != LNX_ADDR_PLAIN || p.cmap.group > 0) return false;
    if (p.act != LNX_ACT_NONE || p.gamma || p.rowscale || p.aux || p.C2) return false;
    if (p.res && !out_f32) return false;  // the residual is fp32 like the output it is added to
    if (out_f32 && ((uintptr_t)p.C & 15) != 0) return false;
    return true;
}

int launch_nt_skinny(const GemmP& p, bool out_f32, hipStream_t st) {
    const dim3 grid(cdiv(p.N, 32), cdiv(p.M, 32));
    if (out_f32) hipLaunchKernelGGL((gemm_nt_skinny_kernel<true>), grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL((gemm_nt_skinny_kernel<false>), grid, dim3(64), 0, st, p);
    return 0;
}

}  // namespace lnxg
