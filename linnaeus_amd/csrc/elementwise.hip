// Small HBM-bound data-movement kernels of the mFormerV1 path (gfx950).
// All are grid-stride, 16-byte vectorised where the layout allows, fp32 math.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

inline int ew_grid(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------
// stem im2col: NCHW fp32 image -> [B*Ho*Wo, ldp] patches, k = c*16 + kh*4 + kw
// one thread per (patch row, channel, kh): reads 4 contiguous floats, writes 4 T
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int Cin, int H, int W, int ldp) {
    const int Ho = H >> 2, Wo = W >> 2;
    const int units_per_row = ldp >> 2;  // groups of 4 columns (incl. zero padding)
    const int64_t total = (int64_t)B * Ho * Wo * units_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % units_per_row);
        const int64_t row = i / units_per_row;
        const int wo = (int)(row % Wo);
        const int64_t t = row / Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < Cin * 4) {
            const int c = u >> 2, kh = u & 3;
            v = *reinterpret_cast<const float4*>(x + (((int64_t)b * Cin + c) * H + 4 * ho + kh) * W + 4 * wo);
        }
        T* o = out + row * ldp + 4 * u;
        o[0] = from_f<T>(v.x);
        o[1] = from_f<T>(v.y);
        o[2] = from_f<T>(v.z);
        o[3] = from_f<T>(v.w);
    }
}

// out[m, c] = rowscale[m / rps] * in[map(m), c]
template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ in, int64_t ldin, RowMap map, const float* __restrict__ rowscale,
                                                         int rps, T* __restrict__ out, int64_t ldout, int M, int C) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)M * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / c4n);
        const int c = (int)(i % c4n) * 4;
        const float s = rowscale ? rowscale[m / rps] : 1.f;
        const float4 v = *reinterpret_cast<const float4*>(in + map_row(map, m) * ldin + c);
        T* o = out + (int64_t)m * ldout + c;
        o[0] = from_f<T>(v.x * s);
        o[1] = from_f<T>(v.y * s);
        o[2] = from_f<T>(v.z * s);
        o[3] = from_f<T>(v.w * s);
    }
}

// dz = s * gamma * g ; dgamma += sum_m s * g * z.  Block = 256 threads as (rows x C/4
// column groups); per-thread column partials are reduced through LDS, one atomic per column.
template <typename T>
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float* __restrict__ g, const T* __restrict__ z, const float* __restrict__ gamma,
                                                             const float* __restrict__ rowscale, int rps, T* __restrict__ dz,
                                                             float* __restrict__ dgamma, int M, int C) {
    extern __shared__ float red[];  // [rows_per_block][C]
    const int c4n = C >> 2;
    const int rows_pb = 256 / c4n > 0 ? 256 / c4n : 1;
    const int tr = threadIdx.x / c4n, tc = threadIdx.x % c4n;
    const bool active = tr < rows_pb;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) gm = *reinterpret_cast<const float4*>(gamma + 4 * tc);
    if (active) {
        for (int m = blockIdx.x * rows_pb + tr; m < M; m += gridDim.x * rows_pb) {
            const float s = rowscale ? rowscale[m / rps] : 1.f;
            const float4 gv = *reinterpret_cast<const float4*>(g + (int64_t)m * C + 4 * tc);
            const T* zp = z + (int64_t)m * C + 4 * tc;
            const float z0 = to_f(zp[0]), z1 = to_f(zp[1]), z2 = to_f(zp[2]), z3 = to_f(zp[3]);
            T* o = dz + (int64_t)m * C + 4 * tc;
            o[0] = from_f<T>(s * gm.x * gv.x);
            o[1] = from_f<T>(s * gm.y * gv.y);
            o[2] = from_f<T>(s * gm.z * gv.z);
            o[3] = from_f<T>(s * gm.w * gv.w);
            acc.x += s * gv.x * z0;
            acc.y += s * gv.y * z1;
            acc.z += s * gv.z * z2;
            acc.w += s * gv.w * z3;
        }
        *reinterpret_cast<float4*>(red + tr * C + 4 * tc) = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int r = 0; r < rows_pb; ++r) t += red[r * C + c];
        atomicAdd(dgamma + c, t);
    }
}

__global__ __launch_bounds__(256) void fill_rows_kernel(const float* __restrict__ vec, float* __restrict__ out, int64_t ldout, RowMap map, int M, int C) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)M * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / c4n);
        const int c = (int)(i % c4n) * 4;
        *reinterpret_cast<float4*>(out + map_row(map, m) * ldout + c) = *reinterpret_cast<const float4*>(vec + c);
    }
}

// out[c] += sum_m in[map(m), c]; block handles a strip of rows, threads over columns
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ in, int64_t ldin, RowMap map, float* __restrict__ out, int M, int C,
                                                          int rows_per_block) {
    const int m0 = blockIdx.x * rows_per_block;
    const int m1 = min(M, m0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int m = m0; m < m1; ++m) t += in[map_row(map, m) * ldin + c];
        atomicAdd(out + c, t);
    }
}

__global__ __launch_bounds__(256) void agg2_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w2,
                                                       const float* __restrict__ bias1, float* __restrict__ out, int64_t total) {
    const float w0 = w2[0], w1 = w2[1], bb = bias1[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) out[i] = w0 * a[i] + w1 * b[i] + bb;
}

__global__ __launch_bounds__(256) void agg2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ w2, float* __restrict__ da, float* __restrict__ db_, float* __restrict__ dw2,
                                                       float* __restrict__ dbias1, int64_t total) {
    __shared__ float red[3][4];
    const float w0 = w2[0], w1 = w2[1];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = dout[i];
        da[i] = w0 * d;
        db_[i] = w1 * d;
        s0 += d * a[i];
        s1 += d * b[i];
        s2 += d;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
        red[2][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dw2 + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(dw2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        atomicAdd(dbias1, red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_meta_kernel(const float* __restrict__ meta, int width, int off, int dim, T* __restrict__ out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 16) return;
    const int b = i >> 4, j = i & 15;
    out[i] = from_f<T>(j < dim ? meta[(int64_t)b * width + off + j] : 0.f);
}

// ---------------------------------------------------------------------------------
// per-step parameter preparation (descriptor table on the device)
// ---------------------------------------------------------------------------------
constexpr int PREP_ELEMS = 2048;  // destination elements per workgroup

constexpr int PREP_TR = 32, PREP_TK = 64;  // transposed copy: source tile of 32 rows x 64 columns per workgroup

__host__ __device__ inline int prep_main_blocks(int rows, int ld) { return (int)(((int64_t)rows * ld + PREP_ELEMS - 1) / PREP_ELEMS); }

template <typename T>
__global__ __launch_bounds__(256) void prep_weights_kernel(const lnx_prep_desc* __restrict__ descs, int ndesc) {
    __shared__ float tile[PREP_TK][PREP_TR + 1];
    // binary search: last descriptor with block_start <= blockIdx.x
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const lnx_prep_desc d = descs[lo];
    const int lb = blockIdx.x - d.block_start;
    const int64_t n_main = (int64_t)d.rows * d.ld;
    const int64_t base = (int64_t)lb * PREP_ELEMS;
    if (d.mode == LNX_PREP_DW49) {
        // src [C, 49] -> dst fp32 [49][C];  rows = C, cols = 49, ld = C
        float* dst = reinterpret_cast<float*>(d.dst);
        const int64_t n = (int64_t)49 * d.rows;
        for (int64_t i = base + threadIdx.x; i < base + PREP_ELEMS && i < n; i += 256) {
            const int tap = (int)(i / d.rows), c = (int)(i % d.rows);
            dst[i] = d.src[(int64_t)c * 49 + tap];
        }
        return;
    }
    const int cols_out = d.cols;  // logical K of the operand
    const int Cc = d.mode == LNX_PREP_CONV_PERM ? d.cols / d.P : 1;
    auto src_col = [&](int k) {  // CONV_PERM: k = p*C + c  <-  src column c*P + p
        if (d.mode != LNX_PREP_CONV_PERM) return k;
        const int pp = k / Cc;
        return (k - pp * Cc) * d.P + pp;
    };
    const int main_blocks = prep_main_blocks(d.rows, d.ld);
    if (lb < main_blocks) {
        // unpadded, unpermuted rows (every Linear weight): a flat array -- 16-byte loads, four elements a lane, no divisions
        // (element-wise with i / ld and i % ld by run-time values the xl refresh took 1.27 ms)
        if (d.mode != LNX_PREP_CONV_PERM && d.ld == d.cols && ((reinterpret_cast<uintptr_t>(d.src) | reinterpret_cast<uintptr_t>(d.dst)) & 15) == 0 && (n_main & 3) == 0) {
            const int64_t lim = min(n_main, base + PREP_ELEMS);
            for (int64_t i = base + 4 * threadIdx.x; i + 3 < lim; i += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(d.src + i);
                T* o = reinterpret_cast<T*>(d.dst) + i;
                if constexpr (sizeof(T) == 2) {
                    uint2 pk;
                    T* hh = reinterpret_cast<T*>(&pk);
                    hh[0] = from_f<T>(v.x); hh[1] = from_f<T>(v.y); hh[2] = from_f<T>(v.z); hh[3] = from_f<T>(v.w);
                    *reinterpret_cast<uint2*>(o) = pk;
                } else {
                    *reinterpret_cast<float4*>(o) = v;
                }
            }
            return;
        }
        for (int64_t i = base + threadIdx.x; i < base + PREP_ELEMS && i < n_main; i += 256) {
            const int r = (int)(i / d.ld), k = (int)(i % d.ld);
            const float v = k < cols_out ? d.src[(int64_t)r * d.cols + src_col(k)] : 0.f;
            reinterpret_cast<T*>(d.dst)[i] = from_f<T>(v);
        }
        return;
    }
    if (d.dst_t == nullptr) return;
    // Transposed copy through an LDS tile: rows of the source are read along k (coalesced), rows of the destination
    // written along r (coalesced).  Only the [cols_out, rows] block is written (several tensors may share one
    // row-padded destination, so padding columns are left as the caller zeroed them).
    const int tiles_k = (cols_out + PREP_TK - 1) / PREP_TK;
    const int tt = lb - main_blocks;
    const int r0 = (tt / tiles_k) * PREP_TR, k0 = (tt % tiles_k) * PREP_TK;
    for (int e = threadIdx.x; e < PREP_TR * PREP_TK; e += 256) {
        const int rr = e / PREP_TK, kk = e % PREP_TK;
        const int r = r0 + rr, k = k0 + kk;
        tile[kk][rr] = (r < d.rows && k < cols_out) ? d.src[(int64_t)r * d.cols + src_col(k)] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PREP_TR * PREP_TK; e += 256) {
        const int kk = e / PREP_TR, rr = e % PREP_TR;
        const int r = r0 + rr, k = k0 + kk;
        if (r < d.rows && k < cols_out) reinterpret_cast<T*>(d.dst_t)[(int64_t)k * d.ld_t + r] = from_f<T>(tile[kk][rr]);
    }
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
    do {                                               \
        if ((dtype) == LNX_BF16) {                     \
            typedef bf16_t T;                          \
            __VA_ARGS__;                               \
        } else if ((dtype) == LNX_F32) {               \
            typedef float T;                           \
            __VA_ARGS__;                               \
        } else {                                       \
            LNX_CHECK(false, "bad dtype %d", (dtype)); \
        }                                              \
    } while (0)

extern "C" int lnx_im2col_stem(const float* x, int B, int Cin, int H, int W, void* patches, int dtype, int ldp, void* stream) {
    LNX_CHECK(x && patches, "lnx_im2col_stem: null operand");
    LNX_CHECK(H % 4 == 0 && W % 4 == 0 && ldp % 4 == 0 && ldp >= Cin * 16, "lnx_im2col_stem: bad geometry H=%d W=%d ldp=%d", H, W, ldp);
    const int64_t total = (int64_t)B * (H / 4) * (W / 4) * (ldp / 4);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((im2col_stem_kernel<T>), dim3(ew_grid(total, 256)), dim3(256), 0, st, x, (T*)patches, B, Cin, H, W, ldp));
    LNX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// dropout with a caller-drawn keep mask (lnx_dropout_mul / lnx_dropout_residual): 8 elements per thread
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void dropout_mul_kernel(T* __restrict__ x, const unsigned char* __restrict__ mask, float inv_keep, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const uint2 mk = *reinterpret_cast<const uint2*>(mask + i * 8);
        const unsigned char* mb = reinterpret_cast<const unsigned char*>(&mk);
        T* px = x + i * 8;
        float v[8];
        if constexpr (sizeof(T) == 2) {
            Vec16<T> t;
            t.raw = ld16(px);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mb[j] ? t.get(j) * inv_keep : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t.set(j, v[j]);
            st16(px, t.raw);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 t = *reinterpret_cast<const float4*>(px + 4 * h);
                t.x = mb[4 * h] ? t.x * inv_keep : 0.f; t.y = mb[4 * h + 1] ? t.y * inv_keep : 0.f;
                t.z = mb[4 * h + 2] ? t.z * inv_keep : 0.f; t.w = mb[4 * h + 3] ? t.w * inv_keep : 0.f;
                *reinterpret_cast<float4*>(px + 4 * h) = t;
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_residual_kernel(const T* __restrict__ z, const unsigned char* __restrict__ mask, float inv_keep,
                                                               const float* __restrict__ rowscale, int rps, const float* __restrict__ res,
                                                               float* __restrict__ out, int C8, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C8;
        const float rs = (rowscale ? rowscale[m / rps] : 1.0f) * inv_keep;
        const uint2 mk = *reinterpret_cast<const uint2*>(mask + i * 8);
        const unsigned char* mb = reinterpret_cast<const unsigned char*>(&mk);
        float zv[8];
        if constexpr (sizeof(T) == 2) {
            Vec16<T> t;
            t.raw = ld16(z + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) zv[j] = t.get(j);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 t = *reinterpret_cast<const float4*>(z + i * 8 + 4 * h);
                zv[4 * h] = t.x; zv[4 * h + 1] = t.y; zv[4 * h + 2] = t.z; zv[4 * h + 3] = t.w;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 r = *reinterpret_cast<const float4*>(res + i * 8 + 4 * h);
            r.x += mb[4 * h] ? zv[4 * h] * rs : 0.f; r.y += mb[4 * h + 1] ? zv[4 * h + 1] * rs : 0.f;
            r.z += mb[4 * h + 2] ? zv[4 * h + 2] * rs : 0.f; r.w += mb[4 * h + 3] ? zv[4 * h + 3] * rs : 0.f;
            *reinterpret_cast<float4*>(out + i * 8 + 4 * h) = r;
        }
    }
}
}  // namespace

extern "C" int lnx_dropout_mul(void* x, int dtype, const unsigned char* mask, float inv_keep, int M, int C, void* stream) {
    LNX_CHECK(x && mask && M > 0 && C > 0 && C % 8 == 0, "lnx_dropout_mul: null operand / C %% 8 != 0");
    LNX_CHECK(dtype == LNX_F32 || dtype == LNX_BF16, "lnx_dropout_mul: bad dtype %d", dtype);
    LNX_CHECK((((uintptr_t)x) & 15) == 0 && (((uintptr_t)mask) & 7) == 0, "lnx_dropout_mul: x 16-byte, mask 8-byte aligned");
    const int64_t n8 = (int64_t)M * C / 8;
    int grid = (int)((n8 + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (dtype == LNX_BF16) hipLaunchKernelGGL(dropout_mul_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, mask, inv_keep, n8);
    else hipLaunchKernelGGL(dropout_mul_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)x, mask, inv_keep, n8);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_dropout_residual(const void* z, int z_dtype, const unsigned char* mask, float inv_keep, const float* rowscale, int rows_per_sample,
                                    const float* res, float* out, int M, int C, void* stream) {
    LNX_CHECK(z && mask && res && out && M > 0 && C > 0 && C % 8 == 0, "lnx_dropout_residual: null operand / C %% 8 != 0");
    LNX_CHECK(z_dtype == LNX_F32 || z_dtype == LNX_BF16, "lnx_dropout_residual: bad dtype %d", z_dtype);
    LNX_CHECK(((((uintptr_t)z) | ((uintptr_t)res) | ((uintptr_t)out)) & 15) == 0 && (((uintptr_t)mask) & 7) == 0, "lnx_dropout_residual: misaligned operand");
    if (rowscale) LNX_CHECK(rows_per_sample > 0, "lnx_dropout_residual: rowscale needs rows_per_sample");
    const int64_t n8 = (int64_t)M * C / 8;
    int grid = (int)((n8 + 255) / 256);
    if (grid > 8192) grid = 8192;
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    if (z_dtype == LNX_BF16)
        hipLaunchKernelGGL(dropout_residual_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, mask, inv_keep, rowscale, rps, res, out, C / 8, n8);
    else
        hipLaunchKernelGGL(dropout_residual_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z, mask, inv_keep, rowscale, rps, res, out, C / 8, n8);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_scale_cast(const float* in, int64_t ldin, lnx_rowmap in_map, const float* rowscale, int rows_per_sample, void* out,
                              int out_dtype, int64_t ldout, int M, int C, void* stream) {
    LNX_CHECK(in && out && M > 0 && C > 0 && C % 4 == 0 && ldin % 4 == 0, "lnx_scale_cast: bad arguments M=%d C=%d", M, C);
    if (rowscale) LNX_CHECK(rows_per_sample > 0, "lnx_scale_cast: rowscale needs rows_per_sample");
    const RowMap map{in_map.group, in_map.pad, in_map.off};
    hipStream_t st = (hipStream_t)stream;
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    DISPATCH_T(out_dtype, hipLaunchKernelGGL((scale_cast_kernel<T>), dim3(ew_grid((int64_t)M * (C / 4), 256)), dim3(256), 0, st, in, ldin, map, rowscale,
                                             rps, (T*)out, ldout, M, C));
    LNX_LAUNCH_CHECK();
    return 0;
}

// dw[c, :] += gamma[c] s[c, :], db[c] += gamma[c] t[c], dgamma[c] += sum_k w[c, k] s[c, k] + b[c] t[c]: one workgroup per channel
// (include/lnx.h: the LayerScale gradient without the saved z)
__global__ __launch_bounds__(256) void layerscale_apply_wgrad_kernel(const float* __restrict__ sp, const float* __restrict__ tp, int64_t lds, const float* __restrict__ w,
                                                                     const float* __restrict__ b, int64_t ldw, const float* __restrict__ gamma, float* __restrict__ dw,
                                                                     float* __restrict__ db, int64_t lddw, float* __restrict__ dgamma, int K) {
    __shared__ float part[4];
    const int c = blockIdx.x;
    const float gm = gamma[c];
    const float* sr = sp + (int64_t)c * lds;
    const float* wr = w + (int64_t)c * ldw;
    float* dr = dw + (int64_t)c * lddw;
    float acc = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const float4 sv = *reinterpret_cast<const float4*>(sr + k), wv = *reinterpret_cast<const float4*>(wr + k);
        float4 d = *reinterpret_cast<const float4*>(dr + k);
        acc += wv.x * sv.x + wv.y * sv.y + wv.z * sv.z + wv.w * sv.w;
        d.x = fmaf(gm, sv.x, d.x);
        d.y = fmaf(gm, sv.y, d.y);
        d.z = fmaf(gm, sv.z, d.z);
        d.w = fmaf(gm, sv.w, d.w);
        *reinterpret_cast<float4*>(dr + k) = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = (part[0] + part[1]) + (part[2] + part[3]);
        if (tp) {
            tot = fmaf(b[c], tp[c], tot);
            db[c] = fmaf(gm, tp[c], db[c]);
        }
        dgamma[c] += tot;
    }
}

extern "C" int lnx_layerscale_apply_wgrad(const float* s, const float* t, int64_t lds, const float* w, const float* b, int64_t ldw, const float* gamma, float* dw, float* db,
                                          int64_t lddw, float* dgamma, int C, int K, void* stream) {
    LNX_CHECK(s && w && gamma && dw && dgamma, "lnx_layerscale_apply_wgrad: null operand");
    LNX_CHECK((t == nullptr) == (b == nullptr) && (t == nullptr) == (db == nullptr), "lnx_layerscale_apply_wgrad: t, b and db are given together or not at all");
    LNX_CHECK(C > 0 && K > 0 && K % 4 == 0 && lds % 4 == 0 && ldw % 4 == 0 && lddw % 4 == 0 && lds >= K && ldw >= K && lddw >= K,
              "lnx_layerscale_apply_wgrad: bad shape C=%d K=%d (K and the leading dimensions are multiples of 4)", C, K);
    LNX_CHECK(((((uintptr_t)s) | ((uintptr_t)w) | ((uintptr_t)dw)) & 15) == 0, "lnx_layerscale_apply_wgrad: s / w / dw must be 16-byte aligned");
    hipLaunchKernelGGL(layerscale_apply_wgrad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, s, t, lds, w, b, ldw, gamma, dw, db, lddw, dgamma, K);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_layerscale_bwd(const float* g, const void* z, int dtype, const float* gamma, const float* rowscale, int rows_per_sample, void* dz,
                                  float* dgamma, int M, int C, void* stream) {
    LNX_CHECK(g && z && gamma && dz && dgamma, "lnx_layerscale_bwd: null operand");
    LNX_CHECK(M > 0 && C > 0 && C % 4 == 0 && C <= 2048, "lnx_layerscale_bwd: bad shape M=%d C=%d", M, C);
    const int c4n = C / 4;
    const int rows_pb = 256 / c4n > 0 ? 256 / c4n : 1;
    LNX_CHECK(c4n <= 256, "lnx_layerscale_bwd: C=%d too large", C);
    int grid = cdiv(M, rows_pb * 8);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    const size_t lds = (size_t)rows_pb * C * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL((layerscale_bwd_kernel<T>), dim3(grid), dim3(256), lds, st, g, (const T*)z, gamma, rowscale, rps, (T*)dz, dgamma, M, C));
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_fill_rows(const float* vec, float* out, int64_t ldout, lnx_rowmap map, int M, int C, void* stream) {
    LNX_CHECK(vec && out && M > 0 && C % 4 == 0, "lnx_fill_rows: bad arguments");
    hipLaunchKernelGGL(fill_rows_kernel, dim3(ew_grid((int64_t)M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, vec, out, ldout,
                       RowMap{map.group, map.pad, map.off}, M, C);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_colsum_rows(const float* in, int64_t ldin, lnx_rowmap map, float* out, int M, int C, void* stream) {
    LNX_CHECK(in && out && M > 0 && C > 0, "lnx_colsum_rows: bad arguments");
    const int rpb = 32;
    hipLaunchKernelGGL(colsum_rows_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, in, ldin, RowMap{map.group, map.pad, map.off}, out, M, C, rpb);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_agg2_fwd(const float* a, const float* b, const float* w2, const float* bias1, float* out, int M, int C, void* stream) {
    LNX_CHECK(a && b && w2 && bias1 && out, "lnx_agg2_fwd: null operand");
    const int64_t total = (int64_t)M * C;
    hipLaunchKernelGGL(agg2_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a, b, w2, bias1, out, total);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_agg2_bwd(const float* dout, const float* a, const float* b, const float* w2, float* da, float* db_, float* dw2, float* dbias1, int M,
                            int C, void* stream) {
    LNX_CHECK(dout && a && b && w2 && da && db_ && dw2 && dbias1, "lnx_agg2_bwd: null operand");
    const int64_t total = (int64_t)M * C;
    int grid = ew_grid(total, 1024);
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(agg2_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, a, b, w2, da, db_, dw2, dbias1, total);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_pack_meta(const float* meta, int width, int off, int dim, void* out, int dtype, int B, void* stream) {
    LNX_CHECK(meta && out && dim > 0 && dim <= 16 && off >= 0 && off + dim <= width, "lnx_pack_meta: bad arguments width=%d off=%d dim=%d", width, off, dim);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((pack_meta_kernel<T>), dim3(cdiv(B * 16, 256)), dim3(256), 0, st, meta, width, off, dim, (T*)out, B));
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_prep_blocks(int rows, int ld, int cols, int ld_t, int has_t) {
    (void)ld_t;
    int n = prep_main_blocks(rows, ld);
    if (has_t) n += ((rows + PREP_TR - 1) / PREP_TR) * ((cols + PREP_TK - 1) / PREP_TK);
    return n;
}

extern "C" int lnx_prep_weights(const lnx_prep_desc* descs_dev, int ndesc, int total_blocks, int dtype, void* stream) {
    LNX_CHECK(descs_dev && ndesc > 0 && total_blocks > 0, "lnx_prep_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL((prep_weights_kernel<T>), dim3(total_blocks), dim3(256), 0, st, descs_dev, ndesc));
    LNX_LAUNCH_CHECK();
    return 0;
}
