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

// Exact-erf GELU on the VALU with as few instructions as possible (the fused conv-MLP kernels and the
// GEMM epilogues of the bf16 mode are bound by VALU issue, not by MFMA or HBM).  erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7):
//   y = 0.5 * erfc(|v|/sqrt2) = 0.5 * (a1 t + ... + a5 t^5) * exp(-v^2/2),  t = 1/(1 + p |v|/sqrt2)
//   Phi(v) = v >= 0 ? 1 - y : y      GELU(v) = v Phi(v) = max(v, 0) - |v y|
// with u = |v| * sqrt(log2(e)/2) so that exp(-v^2/2) = exp2(-u^2) is one v_exp_f32, and 1/x one v_rcp_f32.
// 15 VALU instructions (two of them transcendental) per element.
struct GeluTerms {
    float y;  // 0.5 * erfc(|v|/sqrt2)
    float e;  // exp(-v^2/2)
};
__device__ __forceinline__ GeluTerms gelu_terms(float v) {
    constexpr float K = 0.84932180028801904272f;            // sqrt(log2(e) / 2)
    constexpr float P1 = 0.3275911f * 0.70710678118654752f / K;
    const float u = fabsf(v) * K;
    const float t = __builtin_amdgcn_rcpf(fmaf(P1, u, 1.0f));
    GeluTerms r;
    r.e = __builtin_amdgcn_exp2f(-(u * u));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    r.y = poly * t * r.e;
    return r;
}
__device__ __forceinline__ float gelu_lean(float v) {
    const GeluTerms g = gelu_terms(v);
    return fmaxf(v, 0.f) - fabsf(v * g.y);
}
// act = GELU(v), dgelu = Phi(v) + v phi(v)
__device__ __forceinline__ void gelu_lean_grad(float v, float& act, float& dgelu) {
    const GeluTerms g = gelu_terms(v);
    act = fmaxf(v, 0.f) - fabsf(v * g.y);
    const float cdf = 0.5f + copysignf(0.5f - g.y, v);
    dgelu = fmaf(v * 0.39894228040143267794f, g.e, cdf);
}

// storage-type dispatch: fp32 (strict parity mode) keeps libm erff, bf16 uses the lean form
template <typename T> struct Gelu {
    static __device__ __forceinline__ float fwd(float x) { return gelu_f(x); }
    static __device__ __forceinline__ float grad(float x) { return gelu_grad_f(x); }
};
template <> struct Gelu<bf16_t> {
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
