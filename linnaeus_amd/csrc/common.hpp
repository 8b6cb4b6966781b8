// Shared device/host helpers for the gfx950 kernels of the mFormerV1 path.
// Everything here is written for CDNA4 only: 64-wide wavefronts, MFMA 16x16 tiles,
// 16-byte vector memory operations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LNX_WAVE 64

// ---------------------------------------------------------------------------------
// error reporting (C-ABI: every entry point returns 0 on success, message via
// lnx_last_error())
// ---------------------------------------------------------------------------------
void lnx_set_error(const char* fmt, ...);

#define LNX_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) {                       \
            lnx_set_error(__VA_ARGS__);      \
            return 1;                        \
        }                                    \
    } while (0)

#define LNX_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            lnx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

#define LNX_LAUNCH_CHECK() LNX_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------
// Tile scheduling of the persistent kernels (round 4).  One workgroup per CU that owns the CU's LDS / registers cannot be placed
// beside a resident collective kernel (data-parallel training: the previous backward segment's gradient all-reduce sits on
// 32-64 CUs for hundreds of microseconds); with `tile += gridDim.x` such a workgroup still owes its whole share when it finally
// starts and the launch takes two rounds.  Here every tile -- the first one too -- is DRAWN from an atomic counter: the
// workgroups that run eat the tiles, a latecomer draws a position beyond the end and leaves.
//   g_tile_ctr[slot][x]   counters of one launch; `slot` belongs to the launch's STREAM (api.cpp: tile_slot_of -- a stream's kernels run one
//                         after the other and every kernel leaves its counters at zero, so one set per stream is enough and two
//                         streams never share one; TILE_SLOTS streams per process), one counter per XCD: workgroup b runs on XCD b & 7 (round-robin dispatch), XCD x owns a contiguous eighth of the
//                         tiles and a counter that only its own CUs touch -- the line stays in that XCD's L2 (one counter for the
//                         whole chip bounced between the eight L2s: +10 % on the conv-MLP kernels, 25 000 draws per launch).
//                         Nothing is stolen across XCDs: a collective's workgroups are dealt round-robin too.  Zero at module
//                         load.  A counter sees exactly (tiles + drawers) draws per launch -- every drawer's last draw fails --
//                         so the draw that returns tiles + drawers - 1 is the last, and its owner stores 0 for the next launch.
// LNX_TILE_SCHED=static restores the stride (A/B); LNX_CU_MARGIN / lnx_set_cu_margin() launch on fewer CUs.
// ---------------------------------------------------------------------------------
constexpr int TILE_SLOTS = 64;
static __device__ unsigned g_tile_ctr[TILE_SLOTS][8][32];  // [slot][XCD][0]: one 128-byte line per XCD counter, so that each stays in
                                                           // its own XCD's L2 (one instance per translation unit)

// Draw for kernels whose memory waits are the compiler's (conv-MLP, depthwise): a raw-buffer atomic over a 4-byte buffer -- lane 0 is
// in range, the other lanes fall outside and are dropped by the bounds check: one atomic per wave, no branch, and the compiler counts it
// like any load (its own s_waitcnt vmcnt(N) in front of the first use).  Lane 0 of the result is the counter's previous value; keep
// the first use behind a sched_barrier, or the scheduler hoists it (and its wait) up to the draw.
__device__ __forceinline__ int sched_draw_counted(unsigned* ctr) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ctr, 0, 4, 0x00020000);
    return __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r, (int)(threadIdx.x & 63) * 4, 0, 0);
}
// Draw for kernels that count the in-order memory counter themselves (gemm_nt_v7): inline asm with EXEC narrowed to one lane inside
// the statement, invisible to the compiler's wait insertion; the caller waits before its (inline-asm) consumer reads `result`.
__device__ __forceinline__ void sched_draw(uint32_t& result, unsigned* ctr) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc parses device functions too and knows no "v" registers)
    const uint32_t one = 1u;
    uint64_t save;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                 : "=&v"(result), "=&s"(save) : "v"(ctr), "v"(one) : "memory");
#endif
}
__device__ __forceinline__ void sched_reset(unsigned* ctr) { __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A launch of `grid` workgroups splits its n tiles into nx = min(8, grid) contiguous parts, one per XCD (part = blockIdx.x % nx:
// round-robin dispatch; fewer than 8 workgroups = fewer parts, so that no part is left without a worker).  TileShare: this
// workgroup's part [base, base + cnt), its index among the part's `workers` workgroups, and the counter line the part draws from.
struct TileShare {
    int part, base, cnt, index, workers;
};
__device__ __forceinline__ TileShare tile_share(int n) {
    const int grid = (int)gridDim.x, nx = grid < 8 ? grid : 8;
    TileShare t;
    t.part = (int)blockIdx.x % nx;
    t.index = (int)blockIdx.x / nx;
    t.workers = grid / nx + (t.part < grid % nx ? 1 : 0);
    const int q = n / nx, r = n % nx;
    t.cnt = q + (t.part < r ? 1 : 0);
    t.base = t.part < r ? t.part * (q + 1) : r * (q + 1) + (t.part - r) * q;
    return t;
}

// host side (api.cpp)
int persistent_cus(int cus);   // CUs a persistent launch may occupy: cus - margin
void set_cu_margin(int m);
bool tile_sched_static();      // LNX_TILE_SCHED=static, read per launch
int tile_slot_of(hipStream_t st);  // the counter set launches on this stream draw from (one per stream: a stream's kernels run one after the other); -1 once 64 streams hold one (static stride)
int device_cus();              // cached multiProcessorCount of the current device (0 on failure)
// norm.hip: postpone the second stage of a LayerNorm-style column-sum reduction (part[nwg][2C] -> dw, db) to lnx_layernorm_bwd_flush()
void ln_postpone_reduce(const float* part, int nwg, int C, float* dw, float* db, int slices, hipStream_t st);

// ---------------------------------------------------------------------------------
// scalar type traits.  T is the storage type of activations / GEMM operands:
// bf16 (production) or float (strict-parity mode).  Accumulation is always fp32.
// ---------------------------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int EPV = 4;  // elements per 16-byte vector
    static constexpr int ID = 0;
};
template <> struct TT<bf16_t> {
    static constexpr int EPV = 8;
    static constexpr int ID = 1;
};

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T  <->  floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    uint4 raw;
    __device__ __forceinline__ float get(int i) const { return ((const float*)&raw)[i]; }
    __device__ __forceinline__ void set(int i, float v) { ((float*)&raw)[i] = v; }
};
template <> struct Vec16<bf16_t> {
    uint4 raw;
    __device__ __forceinline__ float get(int i) const { return (float)((const bf16_t*)&raw)[i]; }
    __device__ __forceinline__ void set(int i, float v) { ((bf16_t*)&raw)[i] = (bf16_t)v; }
};

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// exact-erf GELU and its derivative (nn.GELU() default, reference blocks/mlp.py:38)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// erf-GELU on the VALU for the bf16 mode, with as few issue cycles as possible (the fused conv-MLP kernels and the
// GEMM epilogues are bound by VALU issue there, not by MFMA or HBM; transcendental ops are quarter rate, and plain
// FMA chains pair up into v_pk_fma_f32).  erf(x / sqrt2) = xc * P(xc^2) with xc = clamp(x, +-3 sqrt2) and P a degree-8
// near-minimax polynomial: |Phi error| <= 1.4e-5 (1.1e-5 of it the clamp at |x| = 4.24), |GELU error| <= 5.8e-5
// over all x -- far inside bf16 rounding (2^-9 relative) of the values it feeds.  The fp32 mode keeps libm erff.
__device__ __forceinline__ float erf_sqrt2_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.2426405f, 4.2426405f);
    const float t = xc * xc;
    float p = 1.1254853916e-10f;
    p = fmaf(p, t, -1.0744679894e-08f);
    p = fmaf(p, t, 4.5368678889e-07f);
    p = fmaf(p, t, -1.1292854487e-05f);
    p = fmaf(p, t, 1.8718494423e-04f);
    p = fmaf(p, t, -2.2188186466e-03f);
    p = fmaf(p, t, 1.9636284401e-02f);
    p = fmaf(p, t, -1.3269389935e-01f);
    p = fmaf(p, t, 7.9780627149e-01f);
    return p * xc;
}
__device__ __forceinline__ float gelu_lean(float v) {
    const float hv = 0.5f * v;
    return fmaf(hv, erf_sqrt2_poly(v), hv);
}
// act = GELU(v), dgelu = Phi(v) + v phi(v); exp(-v^2/2) = exp2(-(v K)^2) is the one transcendental
__device__ __forceinline__ void gelu_lean_grad(float v, float& act, float& dgelu) {
    constexpr float K = 0.84932180028801904272f;  // sqrt(log2(e) / 2)
    const float cdf = fmaf(0.5f, erf_sqrt2_poly(v), 0.5f);
    const float u = v * K;
    const float e = __builtin_amdgcn_exp2f(-(u * u));
    act = v * cdf;
    dgelu = fmaf(v * 0.39894228040143267794f, e, cdf);
}

// Two-wide forms on ext_vector float2: every multiply / FMA below is one v_pk_mul_f32 / v_pk_fma_f32 for two elements
// (hipcc does not pair the scalar chains by itself inside the fused kernels' unrolled loops: 21 VALU instructions per
// hidden element measured there against ~8 with these).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t splat2(float v) { return f32x2_t{v, v}; }
__device__ __forceinline__ f32x2_t erf_sqrt2_poly2(f32x2_t x) {
    f32x2_t xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -4.2426405f, 4.2426405f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -4.2426405f, 4.2426405f);
    const f32x2_t t = xc * xc;
    f32x2_t p = splat2(1.1254853916e-10f);
    p = p * t + splat2(-1.0744679894e-08f);
    p = p * t + splat2(4.5368678889e-07f);
    p = p * t + splat2(-1.1292854487e-05f);
    p = p * t + splat2(1.8718494423e-04f);
    p = p * t + splat2(-2.2188186466e-03f);
    p = p * t + splat2(1.9636284401e-02f);
    p = p * t + splat2(-1.3269389935e-01f);
    p = p * t + splat2(7.9780627149e-01f);
    return p * xc;
}
__device__ __forceinline__ f32x2_t gelu_lean2(f32x2_t v) {
    const f32x2_t hv = v * splat2(0.5f);
    return hv * erf_sqrt2_poly2(v) + hv;
}
__device__ __forceinline__ void gelu_lean_grad2(f32x2_t v, f32x2_t& act, f32x2_t& dgelu) {
    constexpr float K = 0.84932180028801904272f;  // sqrt(log2(e) / 2)
    const f32x2_t cdf = erf_sqrt2_poly2(v) * splat2(0.5f) + splat2(0.5f);
    const f32x2_t u = v * splat2(K);
    const f32x2_t nuu = -(u * u);
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(nuu.x);
    e.y = __builtin_amdgcn_exp2f(nuu.y);
    act = v * cdf;
    dgelu = (v * splat2(0.39894228040143267794f)) * e + cdf;
}
// ---- N chains in lockstep ------------------------------------------------------------------------------------------
// hipcc emits each Horner chain of the forms above back to back, with an s_nop between every dependent v_pk_fma_f32
// (packed-fp32 result hazard): 11 serial packed ops per element pair, the VALU idling on its own latency.  The _n forms
// advance N independent pairs one polynomial step at a time, so in-order issue always has N-1 independent packed
// operations between two dependent ones.  Same arithmetic per element, bit-identical results.
template <int N>
__device__ __forceinline__ void erf_sqrt2_poly2_n(const f32x2_t (&x)[N], f32x2_t (&r)[N]) {
    constexpr float K[9] = {1.1254853916e-10f, -1.0744679894e-08f, 4.5368678889e-07f, -1.1292854487e-05f, 1.8718494423e-04f,
                            -2.2188186466e-03f, 1.9636284401e-02f, -1.3269389935e-01f, 7.9780627149e-01f};
    f32x2_t xc[N], t[N], p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        xc[i].x = __builtin_amdgcn_fmed3f(x[i].x, -4.2426405f, 4.2426405f);
        xc[i].y = __builtin_amdgcn_fmed3f(x[i].y, -4.2426405f, 4.2426405f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = xc[i] * xc[i];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = splat2(K[0]) * t[i] + splat2(K[1]);
#pragma unroll
    for (int k = 2; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = p[i] * t[i] + splat2(K[k]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = p[i] * xc[i];
}
template <int N>
__device__ __forceinline__ void gelu_lean2_n(const f32x2_t (&v)[N], f32x2_t (&act)[N]) {
    f32x2_t e[N];
    erf_sqrt2_poly2_n<N>(v, e);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const f32x2_t hv = v[i] * splat2(0.5f);
        act[i] = hv * e[i] + hv;
    }
}
template <int N>
__device__ __forceinline__ void gelu_lean_grad2_n(const f32x2_t (&v)[N], f32x2_t (&act)[N], f32x2_t (&dgelu)[N]) {
    constexpr float K = 0.84932180028801904272f;  // sqrt(log2(e) / 2)
    f32x2_t e[N], ex[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const f32x2_t u = v[i] * splat2(K);
        const f32x2_t nuu = -(u * u);
        ex[i].x = __builtin_amdgcn_exp2f(nuu.x);  // transcendental issued first: its latency runs under the polynomial
        ex[i].y = __builtin_amdgcn_exp2f(nuu.y);
    }
    erf_sqrt2_poly2_n<N>(v, e);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const f32x2_t cdf = e[i] * splat2(0.5f) + splat2(0.5f);
        act[i] = v[i] * cdf;
        dgelu[i] = (v[i] * splat2(0.39894228040143267794f)) * ex[i] + cdf;
    }
}
// K accumulator vectors at once: h[k] = GELU(h[k] + b[k])
template <int K>
__device__ __forceinline__ void gelu4_bias_n(f32x4_t* h, const f32x4_t* b) {
    f32x2_t v[2 * K], a[2 * K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[2 * k] = f32x2_t{h[k][0] + b[k][0], h[k][1] + b[k][1]};
        v[2 * k + 1] = f32x2_t{h[k][2] + b[k][2], h[k][3] + b[k][3]};
    }
    gelu_lean2_n<2 * K>(v, a);
#pragma unroll
    for (int k = 0; k < K; ++k) h[k] = f32x4_t{a[2 * k].x, a[2 * k].y, a[2 * k + 1].x, a[2 * k + 1].y};
}
// h[k] = GELU(h[k] + b[k]),  da[k] *= GELU'(h[k] + b[k])
template <int K>
__device__ __forceinline__ void gelu4_bias_grad_n(f32x4_t* h, const f32x4_t* b, f32x4_t* da) {
    f32x2_t v[2 * K], a[2 * K], d[2 * K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[2 * k] = f32x2_t{h[k][0] + b[k][0], h[k][1] + b[k][1]};
        v[2 * k + 1] = f32x2_t{h[k][2] + b[k][2], h[k][3] + b[k][3]};
    }
    gelu_lean_grad2_n<2 * K>(v, a, d);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        h[k] = f32x4_t{a[2 * k].x, a[2 * k].y, a[2 * k + 1].x, a[2 * k + 1].y};
        da[k] = f32x4_t{da[k][0] * d[2 * k].x, da[k][1] * d[2 * k].y, da[k][2] * d[2 * k + 1].x, da[k][3] * d[2 * k + 1].y};
    }
}

// in-place helpers on the 4-element accumulator vectors of the MFMA kernels: h = GELU(h + b) / (act, da *= GELU'(h + b))
__device__ __forceinline__ void gelu4_bias(f32x4_t& h, const f32x4_t& b) {
    const f32x2_t lo = gelu_lean2(f32x2_t{h[0] + b[0], h[1] + b[1]}), hi = gelu_lean2(f32x2_t{h[2] + b[2], h[3] + b[3]});
    h = f32x4_t{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void gelu4_bias_grad(f32x4_t& h, const f32x4_t& b, f32x4_t& da) {
    f32x2_t a0, d0, a1, d1;
    gelu_lean_grad2(f32x2_t{h[0] + b[0], h[1] + b[1]}, a0, d0);
    gelu_lean_grad2(f32x2_t{h[2] + b[2], h[3] + b[3]}, a1, d1);
    h = f32x4_t{a0.x, a0.y, a1.x, a1.y};
    da = f32x4_t{da[0] * d0.x, da[1] * d0.y, da[2] * d1.x, da[3] * d1.y};
}

// storage-type dispatch: fp32 (strict parity mode) keeps libm erff, bf16 uses the lean form
template <typename T> struct Gelu {
    static __device__ __forceinline__ void fwd16(float (&v)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = gelu_f(v[j]);
    }
    static __device__ __forceinline__ void mulgrad16(float (&v)[16], const float (&x)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= gelu_grad_f(x[j]);
    }
    // v <- GELU(v), d <- GELU'(v)
    static __device__ __forceinline__ void fwd_grad16(float (&v)[16], float (&d)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d[j] = gelu_grad_f(v[j]);
            v[j] = gelu_f(v[j]);
        }
    }
    static __device__ __forceinline__ float fwd(float x) { return gelu_f(x); }
    static __device__ __forceinline__ float grad(float x) { return gelu_grad_f(x); }
};
template <> struct Gelu<bf16_t> {
    static __device__ __forceinline__ void fwd16(float (&v)[16]) {
#pragma unroll
        for (int g = 0; g < 16; g += 8) {  // four pairs in lockstep
            f32x2_t x[4], a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = f32x2_t{v[g + 2 * i], v[g + 2 * i + 1]};
            gelu_lean2_n<4>(x, a);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[g + 2 * i] = a[i].x;
                v[g + 2 * i + 1] = a[i].y;
            }
        }
    }
    static __device__ __forceinline__ void mulgrad16(float (&v)[16], const float (&x)[16]) {
#pragma unroll
        for (int g = 0; g < 16; g += 8) {
            f32x2_t xx[4], a[4], d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xx[i] = f32x2_t{x[g + 2 * i], x[g + 2 * i + 1]};
            gelu_lean_grad2_n<4>(xx, a, d);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[g + 2 * i] *= d[i].x;
                v[g + 2 * i + 1] *= d[i].y;
            }
        }
    }
    static __device__ __forceinline__ void fwd_grad16(float (&v)[16], float (&d)[16]) {
#pragma unroll
        for (int g = 0; g < 16; g += 8) {
            f32x2_t xx[4], a[4], dd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xx[i] = f32x2_t{v[g + 2 * i], v[g + 2 * i + 1]};
            gelu_lean_grad2_n<4>(xx, a, dd);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[g + 2 * i] = a[i].x;
                v[g + 2 * i + 1] = a[i].y;
                d[g + 2 * i] = dd[i].x;
                d[g + 2 * i + 1] = dd[i].y;
            }
        }
    }
    static __device__ __forceinline__ float fwd(float x) { return gelu_lean(x); }
    static __device__ __forceinline__ float grad(float x) {
        float act, dg;
        gelu_lean_grad(x, act, dg);
        return dg;
    }
};

// wave-level reductions over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Row map used by several kernels to address token buffers that carry E extra rows per
// sample: phys_row = m + (m / group) * pad + off   (group == 0 -> identity).
struct RowMap {
    int group, pad, off;
};
__device__ __forceinline__ int64_t map_row(const RowMap& rm, int m) {
    return rm.group > 0 ? (int64_t)m + (int64_t)(m / rm.group) * rm.pad + rm.off : (int64_t)m;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
