// Metadata-head chains in one launch per direction (round 5; include/lnx.h: lnx_meta_heads_fwd / lnx_meta_heads_bwd).
//
// What it replaces (reference mFormerV1.py:282-311, normalization/res_norm_layer.py:23-30; per metadata component and RoPE stage):
//     h0 = ReLU(Linear(d, C)(meta[:, off:off+d]));  x = LN0(h0);  tok = x + LN2(ReLU(W2 . LN1(ReLU(W1 . x))))
// -- until round 4 a chain of 20 fp32 GEMM launches on 128x128 tiles (M = batch rows: two to six tiles each, 46-62 us, matrix pipe 2 %
// busy), 15 LayerNorm and 12 weight-gradient launches per step, ~2.6 ms of side-stream kernel time for ~0.3 GFLOP.  Here ONE workgroup
// takes 16 batch rows through the whole chain (the weights, at most 16 MB per matrix, are read from L2 by every workgroup), every head of
// the model in the same launch; the backward is one launch for the data-gradient chain and one for the weight gradients.
//
// Arithmetic: fp32 storage, v_mfma_f32_16x16x4_f32 (an exact fp32 FMA chain), as the chain always ran.  A product tile is computed
// transposed (first operand = 16 weight rows, second = the 16 activation rows), so lane (c = lane & 15, q = lane >> 4) ends with
// out[row c][16 tile + 4 q .. 4 q + 3]: float4 loads / stores along the row, LayerNorm row sums = two lane shuffles + one LDS exchange
// between the eight waves, column sums (LayerNorm weight / bias gradients) = four lane shuffles.  K is walked 16 at a time with a
// permuted order (MFMA j of a chunk takes k = k0 + 4 q + j from both operands), so each operand chunk is ONE float4 per lane.
// Nothing is atomic: per-row-group partial column sums go to scratch and are added in a fixed order by the weight-gradient launch,
// which also owns every `+=` into the gradient arena -- bit-reproducible run to run.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int MH_WAVES = 8, MH_THREADS = 512, MH_ROWS = 16;
constexpr int MH_MAX_HEADS = 4;   // per launch (kernel arguments: 4 x ~300 bytes)

struct FwdBatch {
    lnx_meta_head_args h[MH_MAX_HEADS];
    int n;
};
struct BwdBatch {
    lnx_meta_head_bwd_args h[MH_MAX_HEADS];
    int n;
    int block_start[MH_MAX_HEADS + 1];  // weight-gradient launch: first block of each head's tasks
};

__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// Raw buffer addressing for the operand streams of chain_matmul: a scalar 64-bit base in the descriptor, one 32-bit byte offset per lane
// and a scalar offset for the k position -- instead of a 64-bit address per lane, tile and chunk in flight (which spilled the chunk buffers).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_mh;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mh_rsrc(const void* base) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint64_t u = (uint64_t)hi << 32 | lo;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 mh_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff /* uniform */) {
    // (the whole vector is cast at once: __builtin_bit_cast(float, v[i]) on an element of the ext-vector folded all four lanes onto element 0 and
    // turned the load into a single dword -- hipcc of ROCm 7.2, caught by the op test)
    const f32x4_t g = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return float4{g[0], g[1], g[2], g[3]};
}
__device__ __forceinline__ void stf4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(f32x4_t v) { return float4{v[0], v[1], v[2], v[3]}; }

// sum over the lanes that share a row (c = lane & 15), then over the eight waves: every lane returns the full-row sums of its row
template <int NV>
__device__ __forceinline__ void row_allreduce(float (&v)[NV], float* red /* [NV][MH_WAVES][16] */, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] += __shfl_xor(v[i], 16);
        v[i] += __shfl_xor(v[i], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[(i * MH_WAVES + wave) * 16 + lane] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < MH_WAVES; ++w) s += red[(i * MH_WAVES + w) * 16 + (lane & 15)];
        v[i] = s;
    }
    __syncthreads();
}

// sum over the 16 rows of the workgroup (lanes with equal q): lane c == 0 of each q group ends with the sums
__device__ __forceinline__ float4 col_reduce(float4 v) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        v.x += __shfl_xor(v.x, m);
        v.y += __shfl_xor(v.y, m);
        v.z += __shfl_xor(v.z, m);
        v.w += __shfl_xor(v.w, m);
    }
    return v;
}

// acc[t] (+)= A[16 rows][K] . W[rows 16 (wave + 8 t) .. +15][K]^T for this wave's NT tiles (every wave has the same number: C % 128 == 0).
// A rows beyond the batch are clamped by the caller (arow).  The weights come from HBM / the Infinity Cache behind ~2 us of latency (every
// workgroup of a head streams the same 0.6-4 MB matrix, in step with the others), so D chunks stay in flight per wave: with one chunk of
// prefetch the C = 768 chain took 417 us for 70 us of matrix-core work (round 5).  A chunk is 32 k = TWO float4 per lane and operand, issued
// back to back: lanes q = 0..3 of a row then ask for one whole 128-byte line (with 16-k chunks the second half of every line came a chunk
// later, after the 32 KiB L1 had been flushed by the other waves' 114 KiB per chunk: every line crossed L2 -> CU twice, ~150 GB/s per CU
// wanted of the ~130 it has).  K % (32 D) == 0; the loop body has no condition (the last D chunks are peeled): counted waits.
// chunk0: the k chunk this workgroup starts at (it wraps around), so that the row groups of a head do not ask for the same weight lines at
// the same moment.  (The fp32 sum order then depends on the row group: 1e-7-level differences between rows, none between runs.)
// Measured (round 5, tools/bench_meta.py, the chip otherwise idle): none of the memory-side measures -- chunks in flight 2 -> 3 -> 4, whole
// lines per request, the row groups of a head on one XCD, this stagger -- moved the C = 768 forward off 170 us (matrix-core work: 70 us);
// the first two fixed the 417 us of the first form.  What is left looks like matrix time + streaming time (4.7 MB per workgroup at the
// ~50-70 GB/s a CU pulls from L2) in sequence rather than overlapped; not resolved, and hidden in the step (side stream).
template <int NT, int D>
__device__ __forceinline__ void chain_matmul(f32x4_t (&acc)[NT], const float* __restrict__ A, int64_t lda, int arow, const float* __restrict__ W, int64_t ldw, int K,
                                             int wave, int lane, int chunk0) {
    const int c = lane & 15, q = lane >> 4;
    // (every operand of the chain is far below 2 GiB: C <= 1024 rows of at most 4 KiB, B rows of C floats)
    const __amdgpu_buffer_rsrc_t ra = mh_rsrc(A), rw = mh_rsrc(W);
    const uint32_t aoff = (uint32_t)(arow * lda + 4 * q) * 4u;
    const uint32_t woff = (uint32_t)((wave * 16 + c) * ldw + 4 * q) * 4u;
    const uint32_t tstride = (uint32_t)(MH_WAVES * 16 * ldw) * 4u;
    float4 a_buf[D][2], w_buf[D][NT][2];
    auto load = [&](int d, uint32_t kbytes) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) a_buf[d][h] = mh_ld(ra, aoff + 64u * h, kbytes);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) w_buf[d][t][h] = mh_ld(rw, woff + (uint32_t)t * tstride + 64u * h, kbytes);
    };
    auto use = [&](int d) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float av[4] = {a_buf[d][h].x, a_buf[d][h].y, a_buf[d][h].z, a_buf[d][h].w};
            // j outside t: consecutive MFMAs go to different accumulators (a dependent fp32 MFMA waits 40 cycles for 32 of issue)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float wv[4] = {w_buf[d][t][h].x, w_buf[d][t][h].y, w_buf[d][t][h].z, w_buf[d][t][h].w};
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], av[j], acc[t], 0, 0, 0);
                }
        }
    };
    const int nchunks = K / 32;
    auto kbytes_of = [&](int i) __attribute__((always_inline)) -> uint32_t {  // i-th chunk of this workgroup's walk (uniform)
        int ch = chunk0 + i;
        if (ch >= nchunks) ch -= nchunks;
        return 128u * (uint32_t)ch;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) load(d, kbytes_of(d));
    for (int i0 = D; i0 < nchunks; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            use(d);
            load(d, kbytes_of(i0 + d));
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) use(d);
}

// the K = 16 product in front of the chain (the metadata slice, zero padded): one 16-k chunk
template <int NT>
__device__ __forceinline__ void chain_matmul16(f32x4_t (&acc)[NT], const float* __restrict__ A, int arow, const float* __restrict__ W, int64_t ldw, int wave, int lane) {
    const int c = lane & 15, q = lane >> 4;
    const float4 a = ldf4(A + (int64_t)arow * 16 + 4 * q);
    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 w = ldf4(W + (int64_t)((wave + MH_WAVES * t) * 16 + c) * ldw + 4 * q);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], av[j], acc[t], 0, 0, 0);
    }
}

// One Linear -> ReLU -> LayerNorm step of the chain on registers: acc holds the product; writes h = ReLU(acc + b) (saved for the backward)
// and y = LN(h) (+ skip) and the row statistics.  Two-pass variance (mean first), as lnx_layernorm_fwd.
template <int NT>
__device__ __forceinline__ void relu_ln_store(f32x4_t (&acc)[NT], int wave, int lane, int C, int row, bool valid, const float* __restrict__ bias,
                                              const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, float* __restrict__ H, float* __restrict__ Y,
                                              int64_t y_stride, float* __restrict__ mean_out, float* __restrict__ rstd_out, const float* __restrict__ skip, float* red) {
    const int c = lane & 15, q = lane >> 4;
    float s[1] = {0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
            const int col = (wave + MH_WAVES * t) * 16 + 4 * q;
            const float4 b = ldf4(bias + col);
            acc[t][0] = fmaxf(acc[t][0] + b.x, 0.f);
            acc[t][1] = fmaxf(acc[t][1] + b.y, 0.f);
            acc[t][2] = fmaxf(acc[t][2] + b.z, 0.f);
            acc[t][3] = fmaxf(acc[t][3] + b.w, 0.f);
            if (valid && H) stf4(H + (int64_t)row * C + col, f4(acc[t]));
            s[0] += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        }
    row_allreduce<1>(s, red, wave, lane);
    const float mean = s[0] / (float)C;
    float v[1] = {0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[t][r] - mean;
                v[0] = fmaf(d, d, v[0]);
            }
        }
    row_allreduce<1>(v, red, wave, lane);
    const float rstd = rsqrtf(v[0] / (float)C + eps);
    if (valid && wave == 0 && q == 0 && mean_out) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
            const int col = (wave + MH_WAVES * t) * 16 + 4 * q;
            const float4 g = ldf4(lnw + col), b = ldf4(lnb + col);
            float4 y;
            y.x = fmaf((acc[t][0] - mean) * rstd, g.x, b.x);
            y.y = fmaf((acc[t][1] - mean) * rstd, g.y, b.y);
            y.z = fmaf((acc[t][2] - mean) * rstd, g.z, b.z);
            y.w = fmaf((acc[t][3] - mean) * rstd, g.w, b.w);
            if (skip && valid) {
                const float4 x = ldf4(skip + (int64_t)row * C + col);
                y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
            }
            if (valid) stf4(Y + (int64_t)row * y_stride + col, y);
        }
}

template <int NT>
__global__ __launch_bounds__(MH_THREADS) void meta_chain_fwd_kernel(const FwdBatch bt) {
    constexpr int FD = NT <= 4 ? 4 : (NT == 6 ? 3 : 2);  // 32-k weight chunks in flight per wave (chain_matmul); K / 32 = 4 NT is a multiple of it
    // blockIdx.x = 8 * row group + head slot: blocks that are equal mod 8 share an XCD under the round-robin placement (speed only), so the 16
    // row groups of a head stream its weight matrices through ONE L2 -- the first workgroup to touch a line pays the Infinity-Cache / HBM
    // round trip, the others hit (a CU pulls ~30 GB/s from beyond L2, ~70 GB/s from it: MI355X_MICROARCH.md, indexed rows)
    const int head = blockIdx.x & 7;
    if (head >= bt.n) return;
    const lnx_meta_head_args& a = bt.h[head];
    const int row0 = (blockIdx.x >> 3) * MH_ROWS;
    if (row0 >= a.B) return;
    __shared__ float red[2 * MH_WAVES * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15;
    const int C = a.C;
    const int row = row0 + c;
    const bool valid = row < a.B;
    const int arow = valid ? row : a.B - 1;
    const int chunk0 = (int)(((blockIdx.x >> 3) & 15) * (unsigned)(C / 32)) >> 4;  // row group r starts r / 16 of the way through K
    // t0: this component's metadata columns, zero-padded to 16 (lnx_pack_meta's layout; the weight-gradient launch reads it too)
    if (tid < MH_ROWS * 16) {
        const int r = tid >> 4, d = tid & 15;
        if (row0 + r < a.B) a.t0[(int64_t)(row0 + r) * 16 + d] = d < a.dim ? a.meta[(int64_t)(row0 + r) * a.meta_width + a.off + d] : 0.f;
    }
    __syncthreads();
    f32x4_t acc[NT];
    auto zero = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    // h0 = ReLU(t0 W0^T + b0), x = LN0(h0)
    zero();
    chain_matmul16<NT>(acc, a.t0, arow, a.w0, a.ldw0, wave, lane);
    relu_ln_store<NT>(acc, wave, lane, C, row, valid, a.b0, a.ln0_w, a.ln0_b, a.eps, a.h0, a.x, C, a.m0, a.r0, nullptr, red);
    __syncthreads();  // x is read back (every wave needs whole rows of it) through this CU's cache
    // h1 = ReLU(x W1^T + b1), n1 = LN1(h1)
    zero();
    chain_matmul<NT, FD>(acc, a.x, C, arow, a.w1, a.ldw1, C, wave, lane, chunk0);
    relu_ln_store<NT>(acc, wave, lane, C, row, valid, a.b1, a.ln1_w, a.ln1_b, a.eps, a.h1, a.n1, C, a.m1, a.r1, nullptr, red);
    __syncthreads();
    // h2 = ReLU(n1 W2^T + b2), tok = x + LN2(h2) -> the token row of this sample
    zero();
    chain_matmul<NT, FD>(acc, a.n1, C, arow, a.w2, a.ldw2, C, wave, lane, chunk0);
    relu_ln_store<NT>(acc, wave, lane, C, row, valid, a.b2, a.ln2_w, a.ln2_b, a.eps, a.h2, a.tok + a.tok_row_offset, a.tok_row_stride, a.m2, a.r2, a.x, red);
}

// LayerNorm backward on registers, followed by the ReLU mask of the Linear in front of the LayerNorm: dy[t] (gradient wrt the LayerNorm
// output; zero in rows beyond the batch) -> dp = [h > 0] rstd (dy g - mean(dy g) - xhat mean(dy g xhat)), stored to DP; the workgroup's
// column sums of dy xhat / dy go to part_g / part_b (row group `rg` of [row_groups][C]).
template <int NT>
__device__ __forceinline__ void ln_relu_bwd(f32x4_t (&dy)[NT], int wave, int lane, int C, int row, bool valid, const float* __restrict__ H,
                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in, const float* __restrict__ lnw, float* __restrict__ DP,
                                            float* __restrict__ part_g, float* __restrict__ part_b, int rg, float* red) {
    const int c = lane & 15, q = lane >> 4;
    const float mean = mean_in[row], rstd = rstd_in[row];  // (row is clamped by the caller)
    constexpr bool KEEP = true;
    float4 xh[KEEP ? NT : 1];
    unsigned long long pos = 0ull;  // bit 4 t + r: h > 0 (the ReLU in front of this LayerNorm let the element through)
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
            const int col = (wave + MH_WAVES * t) * 16 + 4 * q;
            const float4 h = ldf4(H + (int64_t)row * C + col);
            const float4 g = ldf4(lnw + col);
            float4& xt = xh[KEEP ? t : 0];
            xt = float4{(h.x - mean) * rstd, (h.y - mean) * rstd, (h.z - mean) * rstd, (h.w - mean) * rstd};
            pos |= (unsigned long long)((h.x > 0.f ? 1u : 0u) | (h.y > 0.f ? 2u : 0u) | (h.z > 0.f ? 4u : 0u) | (h.w > 0.f ? 8u : 0u)) << (4 * t);
            // column partials first (they take dy itself), then dy <- dy g
            float4 pg = float4{dy[t][0] * xt.x, dy[t][1] * xt.y, dy[t][2] * xt.z, dy[t][3] * xt.w};
            float4 pb = f4(dy[t]);
            pg = col_reduce(pg);
            pb = col_reduce(pb);
            if (c == 0) {
                stf4(part_g + (int64_t)rg * C + col, pg);
                stf4(part_b + (int64_t)rg * C + col, pb);
            }
            dy[t][0] *= g.x; dy[t][1] *= g.y; dy[t][2] *= g.z; dy[t][3] *= g.w;
            s[0] += (dy[t][0] + dy[t][1]) + (dy[t][2] + dy[t][3]);
            s[1] += (dy[t][0] * xt.x + dy[t][1] * xt.y) + (dy[t][2] * xt.z + dy[t][3] * xt.w);
            if (!KEEP && (t & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from interleaving all 16 tiles' loads and shuffle chains of a 2048-wide row: 250 registers spilled)
        }
    row_allreduce<2>(s, red, wave, lane);
    const float m1 = s[0] / (float)C, m2 = s[1] / (float)C;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
            const int col = (wave + MH_WAVES * t) * 16 + 4 * q;
            const unsigned mk = (unsigned)(pos >> (4 * t)) & 15u;
            float4 xt;
            if (KEEP) {
                xt = xh[KEEP ? t : 0];
            } else {
                const float4 h = ldf4(H + (int64_t)row * C + col);
                xt = float4{(h.x - mean) * rstd, (h.y - mean) * rstd, (h.z - mean) * rstd, (h.w - mean) * rstd};
            }
            float4 o;
            o.x = (mk & 1u) ? rstd * (dy[t][0] - m1 - xt.x * m2) : 0.f;
            o.y = (mk & 2u) ? rstd * (dy[t][1] - m1 - xt.y * m2) : 0.f;
            o.z = (mk & 4u) ? rstd * (dy[t][2] - m1 - xt.z * m2) : 0.f;
            o.w = (mk & 8u) ? rstd * (dy[t][3] - m1 - xt.w * m2) : 0.f;
            if (valid) stf4(DP + (int64_t)row * C + col, o);
            if (!KEEP && (t & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
}

template <int NT>
__global__ __launch_bounds__(MH_THREADS) void meta_chain_bwd_kernel(const BwdBatch bt) {
    constexpr int BD = NT <= 4 ? 4 : (NT == 6 ? 3 : 2);
    const int head = blockIdx.x & 7;  // (as meta_chain_fwd_kernel: the row groups of a head on one XCD)
    if (head >= bt.n) return;
    const lnx_meta_head_bwd_args& a = bt.h[head];
    const int row0 = (blockIdx.x >> 3) * MH_ROWS;
    if (row0 >= a.B) return;
    __shared__ float red[2 * MH_WAVES * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int C = a.C;
    const int row = row0 + c;
    const bool valid = row < a.B;
    const int arow = valid ? row : a.B - 1;
    const int rg = blockIdx.x >> 3, nrg = (a.B + MH_ROWS - 1) / MH_ROWS;
    const int chunk0 = (int)((rg & 15) * (unsigned)(C / 32)) >> 4;
    float* const part = a.part;  // [6][nrg][C]: dln2_w, dln2_b, dln1_w, dln1_b, dln0_w, dln0_b
    const int64_t ps = (int64_t)nrg * C;
    f32x4_t acc[NT];
    // tok = x + LN2(h2): the token-row gradient reaches LN2's output and, through the skip, x
    auto load_dtok = [&](int t) -> f32x4_t {
        const int col = (wave + MH_WAVES * t) * 16 + 4 * q;
        if (!valid) return f32x4_t{0.f, 0.f, 0.f, 0.f};
        const float4 v = ldf4(a.g + (int64_t)row * a.g_row_stride + a.g_row_offset + col);
        return f32x4_t{v.x, v.y, v.z, v.w};
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = load_dtok(t);
    ln_relu_bwd<NT>(acc, wave, lane, C, arow, valid, a.h2, a.m2, a.r2, a.ln2_w, a.dp2, part + 0 * ps, part + 1 * ps, rg, red);
    __syncthreads();  // dp2 is read back whole-row by every wave
    // d n1 = dp2 . W2 (the transposed copy: w2t[i][o]), then LN1 / ReLU backward
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    chain_matmul<NT, BD>(acc, a.dp2, C, arow, a.w2t, a.ldw2t, C, wave, lane, chunk0);
    if (!valid) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};  // clamped duplicate rows must not reach the column sums
    }
    ln_relu_bwd<NT>(acc, wave, lane, C, arow, valid, a.h1, a.m1, a.r1, a.ln1_w, a.dp1, part + 2 * ps, part + 3 * ps, rg, red);
    __syncthreads();
    // d x = dp1 . W1 + d tok (skip connection of the ResNormLayer), then LN0 / ReLU backward
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = load_dtok(t);
    chain_matmul<NT, BD>(acc, a.dp1, C, arow, a.w1t, a.ldw1t, C, wave, lane, chunk0);
    if (!valid) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    ln_relu_bwd<NT>(acc, wave, lane, C, arow, valid, a.h0, a.m0, a.r0, a.ln0_w, a.dp0, part + 4 * ps, part + 5 * ps, rg, red);
}

// ---- weight gradients: dW[o][i] += sum_m dP[m][o] X[m][i] for the three Linears of each head, their bias gradients (column sums of
// dP) and the fixed-order sum of the LayerNorm column partials.  One wave = one 64 x 64 block of a dW over all batch rows: per four
// rows ONE float4 of dP and ONE of X per lane feed 16 MFMAs (tile (tt, uu) pairs component tt of the dP vector with component uu of
// the X vector: output row o0 + 4 M + tt, column i0 + 4 N + uu).
__device__ __forceinline__ void wgrad_block(const float* __restrict__ DP, int64_t lddp, int o0, const float* __restrict__ X, int64_t ldx, int i0, int xcols, int B,
                                            float* __restrict__ dW, int64_t lddw, int store_cols, int lane, int stagger) {
    const int c = lane & 15, q = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) acc[tt][uu] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool xin = i0 + 4 * c < xcols;  // (the K = 16 first Linear: only lanes c < 4 hold X columns)
    const float* dp = DP + o0 + 4 * c;
    const float* xp = X + i0 + 4 * c;
    auto fetch = [&](int m, float4& d, float4& x) {
        const bool ok = m < B;
        d = ok ? ldf4(dp + (int64_t)m * lddp) : float4{0.f, 0.f, 0.f, 0.f};
        x = (ok && xin) ? ldf4(xp + (int64_t)m * ldx) : float4{0.f, 0.f, 0.f, 0.f};
    };
    constexpr int D = 8;  // row groups of four in flight (the operands come from L2 / HBM: one group of prefetch left this loop latency-bound)
    // the walk over the batch rows starts at a task-dependent group and wraps (`stagger`): tasks that share a dP or X column slab would
    // otherwise ask for the same lines at the same moment (see chain_matmul)
    const int ngroups = (B + 3) / 4;
    const int g0 = (int)(((unsigned)stagger % 16u) * (unsigned)ngroups) >> 4;
    auto group_of = [&](int i) __attribute__((always_inline)) -> int {
        int gi = g0 + i;
        if (gi >= ngroups) gi -= ngroups;
        return gi;
    };
    float4 d_buf[D], x_buf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d < ngroups) fetch(4 * group_of(d) + q, d_buf[d], x_buf[d]);
        else d_buf[d] = x_buf[d] = float4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i0 = 0; i0 < ngroups; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i0 + d < ngroups) {  // (uniform)
                const float dv[4] = {d_buf[d].x, d_buf[d].y, d_buf[d].z, d_buf[d].w}, xv[4] = {x_buf[d].x, x_buf[d].y, x_buf[d].z, x_buf[d].w};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) acc[tt][uu] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[tt], xv[uu], acc[tt][uu], 0, 0, 0);
                if (i0 + d + D < ngroups) fetch(4 * group_of(i0 + d + D) + q, d_buf[d], x_buf[d]);
            }
        }
    }
    // lane holds D[M = 4 q + r][N = c] of tile (tt, uu): dW[o0 + 4 (4 q + r) + tt][i0 + 4 c + uu]
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = o0 + 4 * (4 * q + r) + tt;
            float* dst = dW + (int64_t)o * lddw + i0 + 4 * c;
            if (store_cols < 0) {
                float4 v = ldf4(dst);
                v.x += acc[tt][0][r]; v.y += acc[tt][1][r]; v.z += acc[tt][2][r]; v.w += acc[tt][3][r];
                stf4(dst, v);
            } else {
#pragma unroll
                for (int uu = 0; uu < 4; ++uu)
                    if (i0 + 4 * c + uu < store_cols) dst[uu] += acc[tt][uu][r];
            }
        }
}

// tasks of one head, in blocks of 256 threads (4 waves): [w2: (C/64)^2 wave tasks][w1: (C/64)^2][w0: C/64] then column tasks (one thread
// per column: three bias sums and six LayerNorm partial sums)
__host__ __device__ inline int mh_wave_tasks(int C) { return 2 * (C / 64) * (C / 64) + C / 64; }
__host__ __device__ inline int mh_wgrad_blocks(int C) { return (mh_wave_tasks(C) + 3) / 4 + (C + 255) / 256; }

__global__ __launch_bounds__(256) void meta_chain_wgrad_kernel(const BwdBatch bt) {
    int hd = 0;
#pragma unroll
    for (int i = 1; i < MH_MAX_HEADS; ++i)
        if ((int)blockIdx.x >= bt.block_start[i]) hd = i;
    const lnx_meta_head_bwd_args& a = bt.h[hd];
    const int blk = (int)blockIdx.x - bt.block_start[hd];
    const int C = a.C, B = a.B, nb = C / 64;
    const int wave_blocks = (mh_wave_tasks(C) + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blk < wave_blocks) {
        int task = blk * 4 + wave;
        if (task >= mh_wave_tasks(C)) return;
        if (task < nb * nb) {
            wgrad_block(a.dp2, C, (task / nb) * 64, a.n1, C, (task % nb) * 64, C, B, a.d_w2, C, -1, lane, task / nb + 5 * (task % nb));
        } else if (task < 2 * nb * nb) {
            task -= nb * nb;
            wgrad_block(a.dp1, C, (task / nb) * 64, a.x, C, (task % nb) * 64, C, B, a.d_w1, C, -1, lane, task / nb + 5 * (task % nb) + 8);
        } else {
            task -= 2 * nb * nb;
            wgrad_block(a.dp0, C, task * 64, a.t0, 16, 0, 16, B, a.d_w0, a.dim, a.dim, lane, task);
        }
        return;
    }
    const int col = (blk - wave_blocks) * 256 + tid;
    if (col >= C) return;
    float s2 = 0.f, s1 = 0.f, s0 = 0.f;
    for (int m = 0; m < B; ++m) {
        s2 += a.dp2[(int64_t)m * C + col];
        s1 += a.dp1[(int64_t)m * C + col];
        s0 += a.dp0[(int64_t)m * C + col];
    }
    a.d_b2[col] += s2;
    a.d_b1[col] += s1;
    a.d_b0[col] += s0;
    const int nrg = (B + MH_ROWS - 1) / MH_ROWS;
    float* const outs[6] = {a.d_ln2_w, a.d_ln2_b, a.d_ln1_w, a.d_ln1_b, a.d_ln0_w, a.d_ln0_b};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float s = 0.f;
        for (int g = 0; g < nrg; ++g) s += a.part[((int64_t)k * nrg + g) * C + col];
        outs[k][col] += s;
    }
}

bool nt_supported(int C) {
    // C = 128 NT with NT tiles per wave; wider rows (lg / xl stage 4: C = 1536 / 2048) do not fit the backward's registers (dy + normalised row +
    // chunk buffers of 12-16 tiles spilled 250 registers) and stay on the launch-by-launch chain (plan.cpp: per stage)
    if (C < 128 || C % 128 != 0 || C > 1024) return false;
    const int nt = C / 128;
    return nt <= 4 || nt == 6 || nt == 8;
}

template <typename Batch, typename Args>
int group_end(const Args* heads, int n_heads, int i0) {  // heads [i0, end) share one width (one template instantiation per launch), at most four
    int e = i0 + 1;
    while (e < n_heads && e - i0 < MH_MAX_HEADS && heads[e].C == heads[i0].C) ++e;
    return e;
}

#define MH_DISPATCH_NT(KERNEL, nt, grid, st, bt)                                                                   \
    do {                                                                                                           \
        switch (nt) {                                                                                              \
            case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(MH_THREADS), 0, st, bt); break;                       \
            case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(MH_THREADS), 0, st, bt); break;                       \
            case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(MH_THREADS), 0, st, bt); break;                       \
            case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(MH_THREADS), 0, st, bt); break;                       \
            case 6: hipLaunchKernelGGL(KERNEL<6>, grid, dim3(MH_THREADS), 0, st, bt); break;                       \
            default: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(MH_THREADS), 0, st, bt); break;                      \
        }                                                                                                          \
    } while (0)

}  // namespace

extern "C" int64_t lnx_meta_heads_bwd_part_floats(int B, int C) { return 6 * (int64_t)((B + MH_ROWS - 1) / MH_ROWS) * C; }
extern "C" int lnx_meta_heads_supported(int C) { return nt_supported(C) ? 1 : 0; }

static int check_head_dims(int B, int C, int dim, const char* who) {
    LNX_CHECK(B > 0 && nt_supported(C) && dim >= 1 && dim <= 16,
              "%s: B=%d C=%d dim=%d (C / 128 in {1, 2, 3, 4, 6, 8} -- lnx_meta_heads_supported -- and dim 1..16)", who, B, C, dim);
    return 0;
}

extern "C" int lnx_meta_heads_fwd(const lnx_meta_head_args* heads, int n_heads, void* stream) {
    LNX_CHECK(heads != nullptr && n_heads > 0, "lnx_meta_heads_fwd: no heads");
    for (int i0 = 0; i0 < n_heads;) {
        const int e = group_end<FwdBatch>(heads, n_heads, i0), n = e - i0;
        FwdBatch bt;
        int maxB = 0;
        for (int i = 0; i < n; ++i) {
            const lnx_meta_head_args& a = heads[i0 + i];
            if (check_head_dims(a.B, a.C, a.dim, "lnx_meta_heads_fwd")) return 1;
            LNX_CHECK(a.meta && a.w0 && a.w1 && a.w2 && a.b0 && a.b1 && a.b2 && a.ln0_w && a.ln0_b && a.ln1_w && a.ln1_b && a.ln2_w && a.ln2_b && a.tok,
                      "lnx_meta_heads_fwd: null operand (head %d)", i0 + i);
            LNX_CHECK(a.t0 && a.h0 && a.x && a.h1 && a.n1 && a.h2 && a.m0 && a.r0 && a.m1 && a.r1 && a.m2 && a.r2, "lnx_meta_heads_fwd: null activation buffer (head %d)", i0 + i);
            LNX_CHECK(a.ldw0 >= 16 && a.ldw0 % 4 == 0 && a.ldw1 >= a.C && a.ldw1 % 4 == 0 && a.ldw2 >= a.C && a.ldw2 % 4 == 0 && a.off >= 0 && a.off + a.dim <= a.meta_width,
                      "lnx_meta_heads_fwd: bad leading dimension / metadata slice (head %d)", i0 + i);
            LNX_CHECK(a.tok_row_stride % 4 == 0 && a.tok_row_offset % 4 == 0 && ((((uintptr_t)a.tok) | ((uintptr_t)a.w0) | ((uintptr_t)a.w1) | ((uintptr_t)a.w2)) & 15) == 0,
                      "lnx_meta_heads_fwd: 16-byte alignment (head %d)", i0 + i);
            bt.h[i] = a;
            if (a.B > maxB) maxB = a.B;
        }
        bt.n = n;
        const dim3 grid(8 * ((maxB + MH_ROWS - 1) / MH_ROWS));
        MH_DISPATCH_NT(meta_chain_fwd_kernel, heads[i0].C / 128, grid, (hipStream_t)stream, bt);
        LNX_LAUNCH_CHECK();
        i0 = e;
    }
    return 0;
}

extern "C" int lnx_meta_heads_bwd(const lnx_meta_head_bwd_args* heads, int n_heads, void* stream) {
    LNX_CHECK(heads != nullptr && n_heads > 0, "lnx_meta_heads_bwd: no heads");
    for (int i0 = 0; i0 < n_heads;) {
        const int e = group_end<BwdBatch>(heads, n_heads, i0), n = e - i0;
        BwdBatch bt;
        int maxB = 0, blocks = 0;
        for (int i = 0; i < MH_MAX_HEADS + 1; ++i) bt.block_start[i] = 0x7fffffff;
        for (int i = 0; i < n; ++i) {
            const lnx_meta_head_bwd_args& a = heads[i0 + i];
            if (check_head_dims(a.B, a.C, a.dim, "lnx_meta_heads_bwd")) return 1;
            LNX_CHECK(a.g && a.w1t && a.w2t && a.ln0_w && a.ln1_w && a.ln2_w && a.t0 && a.h0 && a.x && a.h1 && a.n1 && a.h2 && a.m0 && a.r0 && a.m1 && a.r1 && a.m2 && a.r2,
                      "lnx_meta_heads_bwd: null operand (head %d)", i0 + i);
            LNX_CHECK(a.dp2 && a.dp1 && a.dp0 && a.part && a.d_w0 && a.d_b0 && a.d_ln0_w && a.d_ln0_b && a.d_w1 && a.d_b1 && a.d_ln1_w && a.d_ln1_b && a.d_w2 && a.d_b2 && a.d_ln2_w && a.d_ln2_b,
                      "lnx_meta_heads_bwd: null scratch / gradient buffer (head %d)", i0 + i);
            LNX_CHECK(a.ldw1t >= a.C && a.ldw1t % 4 == 0 && a.ldw2t >= a.C && a.ldw2t % 4 == 0 && a.g_row_stride % 4 == 0 && a.g_row_offset % 4 == 0 &&
                          ((((uintptr_t)a.g) | ((uintptr_t)a.w1t) | ((uintptr_t)a.w2t) | ((uintptr_t)a.d_w1) | ((uintptr_t)a.d_w2)) & 15) == 0,
                      "lnx_meta_heads_bwd: leading dimensions / 16-byte alignment (head %d)", i0 + i);
            bt.h[i] = a;
            bt.block_start[i] = blocks;
            blocks += mh_wgrad_blocks(a.C);
            if (a.B > maxB) maxB = a.B;
        }
        bt.n = n;
        const dim3 grid(8 * ((maxB + MH_ROWS - 1) / MH_ROWS));
        MH_DISPATCH_NT(meta_chain_bwd_kernel, heads[i0].C / 128, grid, (hipStream_t)stream, bt);
        LNX_LAUNCH_CHECK();
        hipLaunchKernelGGL(meta_chain_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt);
        LNX_LAUNCH_CHECK();
        i0 = e;
    }
    return 0;
}
