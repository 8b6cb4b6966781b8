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

// acc += A[m0.., :K] . W[n0.., :K]^T for the wave's 32 x 32 tile (rows beyond M / N clamped: what they produce is not stored)
__device__ __forceinline__ void sk_product(f32x4_t (&acc)[2][2], const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, int M, int N, int K, int m0, int n0,
                                           int s, int g) {
    typedef bf16_t T;
    const T* ar[2] = {A + (int64_t)min(m0 + s, M - 1) * lda + 8 * g, A + (int64_t)min(m0 + 16 + s, M - 1) * lda + 8 * g};
    const T* wr[2] = {W + (int64_t)min(n0 + s, N - 1) * ldw + 8 * g, W + (int64_t)min(n0 + 16 + s, N - 1) * ldw + 8 * g};
    const int nks = (K + 31) / 32;
    for (int k0 = 0; k0 < nks; k0 += SK_KU) {
        uint4 af[SK_KU][2], wf[SK_KU][2];
#pragma unroll
        for (int u = 0; u < SK_KU; ++u) {
            // K is a multiple of 8: a lane's chunk lies wholly inside or wholly outside the row.  Outside: read chunk 0
            // (unconditional load) and zero the W fragment, which zeroes the product.
            const int k = (k0 + u) * 32;
            const bool in = k + 8 * g < K;
            const int ko = in ? k : -8 * g;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[u][i] = ld16(ar[i] + ko);
                wf[u][i] = ld16(wr[i] + ko);
            }
        }
#pragma unroll
        for (int u = 0; u < SK_KU; ++u) {
            const bool in = (k0 + u) * 32 + 8 * g < K;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const uint4 w = in ? wf[u][ni] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) Mfma<T>::run(acc[ni][mi], w, af[u][mi]);  // D[n = 4g + r][m = s]
            }
        }
    }
}

// lane (s, g): row m = m0 + 16 mi + s, columns n0 + 16 ni + 4 g .. + 3
template <bool OUT_F32>
__device__ __forceinline__ void sk_store(const f32x4_t (&acc)[2][2], const float* bias, const float* res, int64_t ldres, void* C, int64_t ldc, int M, int N, int m0,
                                         int n0, int s, int g) {
    typedef bf16_t T;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 16 * mi + s;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + 16 * ni + 4 * g;
            if (n >= N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r] + (bias && n + r < N ? bias[n + r] : 0.f);
            if (res) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) v[r] += res[(int64_t)m * ldres + n + r];
            }
            if constexpr (OUT_F32) {
                float* c = reinterpret_cast<float*>(C) + (int64_t)m * ldc + n;
                if (n + 3 < N && (ldc & 3) == 0) {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < N) c[r] = v[r];
                }
            } else {
                T* c = reinterpret_cast<T*>(C) + (int64_t)m * ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) c[r] = (T)v[r];
            }
        }
    }
}

template <bool OUT_F32>
__global__ __launch_bounds__(64) void gemm_nt_skinny_kernel(const GemmP p) {
    const int lane = threadIdx.x, s = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    sk_product(acc, reinterpret_cast<const bf16_t*>(p.A), p.lda, reinterpret_cast<const bf16_t*>(p.W), p.ldw, p.M, p.N, p.K, m0, n0, s, g);
    sk_store<OUT_F32>(acc, p.bias, p.res, p.ldres, p.C, p.ldc, p.M, p.N, m0, n0, s, g);
}

// Several skinny products in ONE launch (lnx_gemm_nt_group): the model's tail issues the same tiny product once per
// classification head, forward and data gradient -- eight launches of ~8 us each on the critical path of every step.
//   accumulate == 0: independent problems (own A / W / C / bias / res / M / N / K); tile columns are numbered across problems
//   accumulate == 1: C = bias + res + sum_j A_j . W_j^T, one accumulator chain over the terms (M, N, C, bias, res of term 0)
struct SkTerm {
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    const float* res;
    void* C;
    int64_t lda, ldw, ldc, ldres;
    int M, N, K, tile0;
};
struct SkGroup {
    SkTerm t[LNX_GEMM_GROUP_MAX];
    int n, accumulate;
};

template <bool OUT_F32>
__global__ __launch_bounds__(64) void gemm_nt_skinny_group_kernel(const SkGroup gp) {
    const int lane = threadIdx.x, s = lane & 15, g = lane >> 4;
    int q = 0, bx = blockIdx.x;
    if (!gp.accumulate) {
        while (q + 1 < gp.n && bx >= gp.t[q + 1].tile0) ++q;
        bx -= gp.t[q].tile0;
    }
    const SkTerm& o = gp.t[q];
    const int n0 = bx * 32, m0 = blockIdx.y * 32;
    if (m0 >= o.M) return;  // (the grid's rows cover the tallest problem)
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (gp.accumulate) {
        for (int j = 0; j < gp.n; ++j) sk_product(acc, gp.t[j].A, gp.t[j].lda, gp.t[j].W, gp.t[j].ldw, o.M, o.N, gp.t[j].K, m0, n0, s, g);
    } else {
        sk_product(acc, o.A, o.lda, o.W, o.ldw, o.M, o.N, o.K, m0, n0, s, g);
    }
    sk_store<OUT_F32>(acc, o.bias, o.res, o.ldres, o.C, o.ldc, o.M, o.N, m0, n0, s, g);
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

int launch_nt_skinny_group(const GemmP* ps, int n, bool accumulate, bool out_f32, hipStream_t st) {
    SkGroup gp{};
    gp.n = n;
    gp.accumulate = accumulate ? 1 : 0;
    int tiles = 0, rows = 0;
    for (int j = 0; j < n; ++j) {
        const GemmP& p = ps[j];
        SkTerm& t = gp.t[j];
        t.A = reinterpret_cast<const bf16_t*>(p.A); t.W = reinterpret_cast<const bf16_t*>(p.W); t.bias = p.bias; t.res = p.res; t.C = p.C;
        t.lda = p.lda; t.ldw = p.ldw; t.ldc = p.ldc; t.ldres = p.ldres; t.M = p.M; t.N = p.N; t.K = p.K;
        t.tile0 = tiles;
        if (!accumulate || j == 0) tiles += cdiv(p.N, 32);
        if (cdiv(p.M, 32) > rows) rows = cdiv(p.M, 32);
    }
    const dim3 grid(tiles, rows);
    if (out_f32) hipLaunchKernelGGL((gemm_nt_skinny_group_kernel<true>), grid, dim3(64), 0, st, gp);
    else hipLaunchKernelGGL((gemm_nt_skinny_group_kernel<false>), grid, dim3(64), 0, st, gp);
    return 0;
}

int launch_nt_skinny(const GemmP& p, bool out_f32, hipStream_t st) {
    const dim3 grid(cdiv(p.N, 32), cdiv(p.M, 32));
    if (out_f32) hipLaunchKernelGGL((gemm_nt_skinny_kernel<true>), grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL((gemm_nt_skinny_kernel<false>), grid, dim3(64), 0, st, p);
    return 0;
}

}  // namespace lnxg
