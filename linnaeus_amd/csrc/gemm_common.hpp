// Shared pieces of the MFMA GEMM kernels (gemm.hip: 128x128 register-staged kernels for both
// storage types; gemm2.hip: 256x128 LDS-DMA pipelined bf16 kernel).
#pragma once
#include "common.hpp"
#include "../../include/lnx.h"

namespace lnxg {

constexpr int TILE = 128;        // rows of each operand tile
constexpr int ROWB = 128;        // bytes of K per tile row
constexpr int TILE_BYTES = TILE * ROWB;

struct PatchGeom {
    int Hin, Win, Cin;
};

struct GemmP {
    const unsigned char* A;
    const unsigned char* W;
    unsigned char* C;
    unsigned char* C2;
    const unsigned char* aux;
    const float* bias;
    const float* gamma;
    const float* rowscale;
    const float* res;
    int64_t lda, ldw, ldc, ldc2, ldaux, ldres;
    int M, N, K;
    int a_mode, c_mode;
    PatchGeom pg;
    RowMap cmap;
    int act;
    int rows_per_sample;
    int tiles_m, tiles_n;
    const float* sa = nullptr;  // fp8 kernels: dequantisation scales of A and W (device scalars)
    const float* sw = nullptr;
    const uint32_t* mxa = nullptr;  // MXFP8 kernels: E8M0 block scales of A / W, [K/128][rows] dwords (byte j = block 4 ks + j)
    const uint32_t* mxw = nullptr;
    unsigned char* C8 = nullptr;    // F_MXOUT: MXFP8 copy of the bf16 output and its block scales [N/128][M][4]
    unsigned char* C8s = nullptr;
    int64_t ldc8 = 0;
    int stagger = 0;  // tools/experiments/gemm_nt_v5: start delay of the workgroup in the odd wave slot, in steps of ~1024 cycles
    int tile_slot = -1;  // persistent kernels: which set of tile counters this launch draws from (-1: static stride)
};

// XCD-aware bijective remap of the linear workgroup id: consecutive logical tiles land on
// the same XCD (blocks b and b+8 share an XCD under round-robin dispatch) so that the
// n-tiles that re-read one A panel hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// 3-bit XOR key of an LDS tile row; chosen so that the permuted fragment reads below
// (rows {0..3,16..19,32..35,48..51}+4i per 16-lane group) are bank-conflict free.
__device__ __forceinline__ int row_key(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mfma<float> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        const float* fa = reinterpret_cast<const float*>(&a);
        const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], acc, 0, 0, 0);
    }
};

// element offset of the 2x2 patch origin of output pixel m in an NHWC tensor
__device__ __forceinline__ int64_t patch_base(const PatchGeom& g, int m) {
    const int Wo = g.Win >> 1, Ho = g.Hin >> 1;
    const int wo = m % Wo;
    const int t = m / Wo;
    const int ho = t % Ho;
    const int b = t / Ho;
    return (((int64_t)b * g.Hin + 2 * ho) * g.Win + 2 * wo) * g.Cin;
}
// offset inside a patch for patch-column k = (kh*2 + kw)*Cin + c
__device__ __forceinline__ int64_t patch_col(const PatchGeom& g, int k) {
    const int two_c = 2 * g.Cin;
    const int kh = k >= two_c ? 1 : 0;  // k < 4 Cin (checked by the host entry points): a compare, not a division by a run-time value
    return (int64_t)kh * g.Win * g.Cin + (k - kh * two_c);
}

// Shared epilogue: the wave's 64x64 sub-tile starts at (mrow0, ncol0); lane (s, g) holds, for each
// mi, 16 consecutive n of row m (see the orientation note in gemm.hip).
template <typename T, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4_t (&acc)[4][4], int mrow0, int ncol0, int lane) {
    constexpr int EPV = TT<T>::EPV;
    const int s = lane & 15, g = lane >> 4;
    const int nb = ncol0 + g * 16;  // first n of this lane
    if (nb >= p.N) return;
    const int nvalid = min(16, p.N - nb);
    // Every load below is unconditional (clamped column / row, uniform branches only): a load under a per-lane branch
    // whose result is merged with a constant is waited for on the spot, which made this prologue 32 + 4 serial memory
    // round trips per workgroup.
    float bias[16], gam[16];
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 16; ++j) bias[j] = p.bias[min(nb + j, p.N - 1)];
    }
    if (p.gamma) {
#pragma unroll
        for (int j = 0; j < 16; ++j) gam[j] = p.gamma[min(nb + j, p.N - 1)];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (!p.bias || j >= nvalid) bias[j] = 0.f;
        if (!p.gamma || j >= nvalid) gam[j] = 1.f;
    }
    // Phase 1: every global load of the epilogue (aux, residual, row scale) for all four row slots, before
    // the first store.  C / C2 may alias res or aux as far as the compiler knows, so loads issued between
    // stores would each wait for the previous store: one memory round trip per row slot.
    const bool has_aux = (p.act == LNX_ACT_GELU_BWD || p.act == LNX_ACT_RELU_BWD || p.act == LNX_ACT_MUL_AUX);
    // one staging buffer: aux when the activation needs it, else the residual (a launch with both loads
    // the residual late, in phase 2)
    const bool res_early = p.res && !has_aux;
    // 16-byte accesses for the staged operand: a kernel-uniform condition (a per-lane one would split the loads)
    const bool vec_aux = has_aux && p.N % 16 == 0 && (((uintptr_t)p.aux) & 15) == 0 && (p.ldaux * sizeof(T)) % 16 == 0;
    const bool vec_res = res_early && p.N % 16 == 0 && (((uintptr_t)p.res) & 15) == 0 && (p.ldres * 4) % 16 == 0 && p.c_mode != LNX_ADDR_PATCH2;
    float av[4][16], rs[4];
    int64_t roffs[4];
    int64_t coffs[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = min(mrow0 + (s >> 2) * 16 + mi * 4 + (s & 3), p.M - 1);  // rows beyond M are loaded, never stored
        rs[mi] = p.rowscale ? p.rowscale[m / p.rows_per_sample] : 1.f;
        if (p.c_mode == LNX_ADDR_PATCH2) {
            coffs[mi] = patch_base(p.pg, m) + patch_col(p.pg, nb);
            roffs[mi] = coffs[mi];
        } else {
            const int64_t row = map_row(p.cmap, m);
            coffs[mi] = row * p.ldc + nb;
            roffs[mi] = row * p.ldres + nb;
        }
        if (has_aux) {
            const T* ax = reinterpret_cast<const T*>(p.aux) + (int64_t)m * p.ldaux + nb;
            if (vec_aux) {
#pragma unroll
                for (int h = 0; h < 16 / EPV; ++h) {
                    Vec16<T> t;
                    t.raw = ld16(ax + h * EPV);
#pragma unroll
                    for (int j = 0; j < EPV; ++j) av[mi][h * EPV + j] = t.get(j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) av[mi][j] = to_f(ax[min(j, nvalid - 1)]);
            }
        } else if (res_early) {
            const float* rp = p.res + roffs[mi];
            if (vec_res) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float4 t = *reinterpret_cast<const float4*>(rp + 4 * h);
                    av[mi][4 * h + 0] = t.x;
                    av[mi][4 * h + 1] = t.y;
                    av[mi][4 * h + 2] = t.z;
                    av[mi][4 * h + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) av[mi][j] = rp[min(j, nvalid - 1)];
            }
        }
    }
    // Phase 2: arithmetic and stores
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mrow0 + (s >> 2) * 16 + mi * 4 + (s & 3);
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[ni * 4 + r] = acc[ni][mi][r] + bias[ni * 4 + r];
        float d2[16];  // what goes to the second output: the pre-activation, or (GELU_D) the derivative of the activation
        if (p.act == LNX_ACT_GELU_D) {
            Gelu<T>::fwd_grad16(v, d2);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) d2[j] = v[j];
        }
        if (p.C2) {
            T* c2 = reinterpret_cast<T*>(p.C2) + (int64_t)m * p.ldc2 + nb;
            if (nvalid == 16 && ((((uintptr_t)c2) & 15) == 0)) {
                Vec16<T> o;
#pragma unroll
                for (int h = 0; h < 16 / EPV; ++h) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) o.set(j, d2[h * EPV + j]);
                    st16(c2 + h * EPV, o.raw);
                }
            } else {
                for (int j = 0; j < nvalid; ++j) c2[j] = from_f<T>(d2[j]);
            }
        }
        if (p.act == LNX_ACT_MUL_AUX) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] *= av[mi][j];
        } else if (p.act == LNX_ACT_GELU) {
            Gelu<T>::fwd16(v);
        } else if (p.act == LNX_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (p.act == LNX_ACT_GELU_BWD) {
            Gelu<T>::mulgrad16(v, av[mi]);
        } else if (p.act == LNX_ACT_RELU_BWD) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = av[mi][j] > 0.f ? v[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= gam[j] * rs[mi];
        if (res_early) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] += av[mi][j];
        } else if (p.res) {
            const float* rp = p.res + roffs[mi];
            for (int j = 0; j < nvalid; ++j) v[j] += rp[j];
        }

        const int64_t coff = coffs[mi];
        if (OUT_F32) {
            float* cp = reinterpret_cast<float*>(p.C) + coff;
            if (nvalid == 16 && ((((uintptr_t)cp) & 15) == 0)) {
#pragma unroll
                for (int h = 0; h < 4; ++h) *reinterpret_cast<float4*>(cp + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            } else {
                for (int j = 0; j < nvalid; ++j) cp[j] = v[j];
            }
        } else {
            T* cp = reinterpret_cast<T*>(p.C) + coff;
            if (nvalid == 16 && ((((uintptr_t)cp) & 15) == 0)) {
                Vec16<T> o;
#pragma unroll
                for (int h = 0; h < 16 / EPV; ++h) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) o.set(j, v[h * EPV + j]);
                    st16(cp + h * EPV, o.raw);
                }
            } else {
                for (int j = 0; j < nvalid; ++j) cp[j] = from_f<T>(v[j]);
            }
        }
    }
}

// Specialised epilogues of the pipelined kernel.  The generic epilogue above decides everything at run time and
// costs ~850 VALU instructions per wave and 64x64 sub-tile (64-bit address products, zero-filled staging, the
// unused gamma / row-scale multiplies, row-map divisions): 25 % of a K = 384 launch.  The forms the model launches
// on large problems are compiled with their feature set F fixed; preconditions, checked by fast_epilogue_mask():
// identity row map, plain C addressing, N % 16 == 0, every row pitch and base pointer 16-byte aligned, all element
// offsets < 2^31.
enum { F_BIAS = 1, F_C2 = 2, F_GELU = 4, F_GELU_BWD = 8, F_RES = 16, F_MXOUT = 32, F_GENERIC = 1 << 10 };

template <typename T, bool OUT_F32, int F>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmP& p, f32x4_t (&acc)[4][4], int mrow0, int ncol0, int lane) {
    constexpr int EPV = TT<T>::EPV;
    const int s = lane & 15, g = lane >> 4;
    const int nb = ncol0 + g * 16;
    if (nb >= p.N) return;
    const int mbase = mrow0 + (s >> 2) * 16 + (s & 3);  // row of slot mi = mbase + 4 mi
    float bias[16];
    if (F & F_BIAS) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float4 t = *reinterpret_cast<const float4*>(p.bias + nb + 4 * h);
            bias[4 * h + 0] = t.x;
            bias[4 * h + 1] = t.y;
            bias[4 * h + 2] = t.z;
            bias[4 * h + 3] = t.w;
        }
    }
    // loads first (see the generic epilogue): aux for GELU_BWD, or the fp32 residual
    float ld[4][16];
    float rs[4];
    if (F & F_GELU_BWD) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = min(mbase + 4 * mi, p.M - 1);
            const T* ax = reinterpret_cast<const T*>(p.aux) + (m * (int)p.ldaux + nb);
#pragma unroll
            for (int h = 0; h < 16 / EPV; ++h) {
                Vec16<T> t;
                t.raw = ld16(ax + h * EPV);
#pragma unroll
                for (int j = 0; j < EPV; ++j) ld[mi][h * EPV + j] = t.get(j);
            }
        }
    }
    if (F & F_RES) {
        // row scale (DropPath): rows of a 64-row sub-tile span at most two samples when rows_per_sample >= 64
        const bool scaled = p.rowscale != nullptr;
        const int q0 = scaled ? mrow0 / p.rows_per_sample : 0;
        const int edge = (q0 + 1) * p.rows_per_sample;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = min(mbase + 4 * mi, p.M - 1);
            rs[mi] = 1.f;
            if (scaled) rs[mi] = p.rows_per_sample >= 64 ? p.rowscale[q0 + (m >= edge ? 1 : 0)] : p.rowscale[m / p.rows_per_sample];
            const float* rp = p.res + (m * (int)p.ldres + nb);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float4 t = *reinterpret_cast<const float4*>(rp + 4 * h);
                ld[mi][4 * h + 0] = t.x;
                ld[mi][4 * h + 1] = t.y;
                ld[mi][4 * h + 2] = t.z;
                ld[mi][4 * h + 3] = t.w;
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mbase + 4 * mi;
        if (m >= p.M) continue;
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[ni * 4 + r] = (F & F_BIAS) ? acc[ni][mi][r] + bias[ni * 4 + r] : acc[ni][mi][r];
        const bool deriv = (F & F_GELU) && (F & F_C2) && p.act == LNX_ACT_GELU_D;  // kernel-uniform: c2 = GELU'(v) instead of v
        if (F & F_C2) {
            float d2[16];
            if (deriv) {
                Gelu<T>::fwd_grad16(v, d2);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) d2[j] = v[j];
            }
            T* c2 = reinterpret_cast<T*>(p.C2) + (m * (int)p.ldc2 + nb);
            Vec16<T> o;
#pragma unroll
            for (int h = 0; h < 16 / EPV; ++h) {
#pragma unroll
                for (int j = 0; j < EPV; ++j) o.set(j, d2[h * EPV + j]);
                st16(c2 + h * EPV, o.raw);
            }
        }
        if ((F & F_GELU) && !deriv) {
            Gelu<T>::fwd16(v);
        }
        if (F & F_GELU_BWD) {
            if (p.act == LNX_ACT_MUL_AUX) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] *= ld[mi][j];
            } else {
                Gelu<T>::mulgrad16(v, ld[mi]);
            }
        }
        if (F & F_RES) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], rs[mi], ld[mi][j]);
        }
        const int coff = m * (int)p.ldc + nb;
        if (OUT_F32) {
            float* cp = reinterpret_cast<float*>(p.C) + coff;
#pragma unroll
            for (int h = 0; h < 4; ++h) *reinterpret_cast<float4*>(cp + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
        } else {
            T* cp = reinterpret_cast<T*>(p.C) + coff;
            Vec16<T> o;
#pragma unroll
            for (int h = 0; h < 16 / EPV; ++h) {
#pragma unroll
                for (int j = 0; j < EPV; ++j) o.set(j, v[h * EPV + j]);
                st16(cp + h * EPV, o.raw);
                if (F & F_MXOUT) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) v[h * EPV + j] = o.get(j);  // the stored (rounded) values are what gets quantised
                }
            }
            if (F & F_MXOUT) {
                // a 32-element block = this lane's 16 columns and those of lane ^ 16 (same row, g ^ 1); N % 32 == 0, so both
                // lanes are here together
                float am = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) am = fmaxf(am, fabsf(v[j]));
                am = fmaxf(am, __shfl_xor(am, 16, 64));
                const uint32_t bits = __float_as_uint(am);
                int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                e = e < 0 ? 0 : (e > 254 ? 254 : e);
                const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
                uint32_t w[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * h] * inv, -448.f), 448.f), fminf(fmaxf(v[4 * h + 1] * inv, -448.f), 448.f), 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * h + 2] * inv, -448.f), 448.f), fminf(fmaxf(v[4 * h + 3] * inv, -448.f), 448.f), pk, true);
                    w[h] = (uint32_t)pk;
                }
                st16(p.C8 + (int64_t)m * p.ldc8 + nb, make_uint4(w[0], w[1], w[2], w[3]));
                const int kb = nb >> 5;
                if ((g & 1) == 0) p.C8s[((int64_t)(kb >> 2) * p.M + m) * 4 + (kb & 3)] = (unsigned char)e;
            }
        }
    }
}

// feature set of a launch if one of the specialised epilogues covers it, else F_GENERIC
static inline int fast_epilogue_mask(const GemmP& p, bool out_f32) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int esz = 2;  // bf16 kernels only
    if (p.c_mode != LNX_ADDR_PLAIN || p.cmap.group > 0 || p.cmap.off != 0 || p.gamma) return F_GENERIC;
    if (p.N % 16 != 0 || !al16(p.C) || (p.ldc * (out_f32 ? 4 : esz)) % 16 != 0) return F_GENERIC;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)p.M * p.ldc >= lim) return F_GENERIC;
    int f = 0;
    if (p.bias) {
        if (!al16(p.bias)) return F_GENERIC;
        f |= F_BIAS;
    }
    if (p.C2) {
        if (!al16(p.C2) || (p.ldc2 * esz) % 16 != 0 || (int64_t)p.M * p.ldc2 >= lim) return F_GENERIC;
        f |= F_C2;
    }
    if (p.act == LNX_ACT_GELU) f |= F_GELU;
    else if (p.act == LNX_ACT_GELU_D) {
        if (!p.C2) return F_GENERIC;
        f |= F_GELU;
    } else if (p.act == LNX_ACT_GELU_BWD || p.act == LNX_ACT_MUL_AUX) {
        if (!al16(p.aux) || (p.ldaux * esz) % 16 != 0 || (int64_t)p.M * p.ldaux >= lim) return F_GENERIC;
        f |= F_GELU_BWD;
    } else if (p.act != LNX_ACT_NONE) return F_GENERIC;
    if (p.res) {
        if (!al16(p.res) || (p.ldres * 4) % 16 != 0 || (int64_t)p.M * p.ldres >= lim) return F_GENERIC;
        f |= F_RES;
    }
    if (p.rowscale && !p.res) return F_GENERIC;
    // the compiled forms
    if (!out_f32 && (f == 0 || f == F_BIAS || f == (F_BIAS | F_C2 | F_GELU) || f == (F_BIAS | F_GELU) || f == F_GELU_BWD)) return f;  // (bias + GELU alone: an inference plan's fc1)
    if (out_f32 && f == (F_BIAS | F_RES)) return f;
    return F_GENERIC;
}

struct WgradP {
    const unsigned char* dY;
    const unsigned char* A;
    float* dW;
    float* db;
    int64_t lddy, lda, lddw;
    int M, N, K;
    int a_mode;
    PatchGeom pg;
    int k_perm_c, k_store;
    float* ws;  // split-K partial tiles (gemm2.hip) or nullptr = atomics
    int64_t ws_floats;
    int tiles_n, tiles_k, splits, m_per_split;
};

// gemm3.hip: persistent 256x128 kernel whose epilogue stores / fetches ride under the next tile's K loop
bool nt_v7_ok(const GemmP& p, int f, bool out_f32);
int launch_nt_v7(const GemmP& p, int f, bool out_f32, hipStream_t st);

// gemm5.hip: persistent 256x256 kernel (the v4 K loop, tiles drawn from the counters, ring never drained)
bool nt_v9_ok(const GemmP& p, int f, bool out_f32);
int launch_nt_v9(const GemmP& p, int f, bool out_f32, hipStream_t st);

// measurement kernels live outside the product (tools/experiments/); their library registers a dispatcher here.  It returns 0
// when it has launched the product, anything else to decline.  nullptr in the shipped library.
typedef int (*nt_experiment_fn)(const GemmP& p, int f, bool out_f32, hipStream_t st);
extern nt_experiment_fn g_nt_experiment;

// which NT kernel family a launch took (lnx_last_nt_kernel / lnx_nt_kernel_launches: the tests' proof of dispatch)
void note_nt_kernel(int kind);

// gemm_skinny.hip: M <= 256 (one wave per 32x32 output tile, operands straight from L2)
bool nt_skinny_ok(const GemmP& p, int dtype, bool out_f32);
int launch_nt_skinny(const GemmP& p, bool out_f32, hipStream_t st);
int launch_nt_skinny_group(const GemmP* ps, int n, bool accumulate, bool out_f32, hipStream_t st);  // lnx_gemm_nt_group

// gemm2.hip: 256x128 LDS-DMA pipelined kernels (bf16)
int launch_nt_v2(const GemmP& p, bool out_f32, hipStream_t st);
int nt_v2_family(const GemmP& p, bool out_f32, int* f_out);  // V7 / V9 / V4 / V2: the choice launch_nt_v2 makes, without launching
bool nt_v2_ok(const GemmP& p, int dtype);
int launch_tn_v2(const WgradP& p, int splits_hint, hipStream_t st, bool defer);
int tn_flush(hipStream_t st);  // launches the deferred second stages of this thread, if any (st: their stream, or nullptr); 1 = they belong to another stream
int tn_discard();               // forgets them without launching (error paths / teardown); returns how many
bool tn_v2_ok(const WgradP& p, int dtype);

}  // namespace lnxg
using namespace lnxg;
