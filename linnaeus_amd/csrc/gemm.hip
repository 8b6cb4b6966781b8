// MFMA GEMM kernels for gfx950.
//
//  gemm_nt : C[M,N] = epilogue(A[M,K] . W[N,K]^T)   forward + data-gradient GEMMs
//  gemm_tn : dW[N,K] += dY[M,N]^T . A[M,K]          weight gradients (split over M)
//
// Tiling (both kernels): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2,
// each wave owns a 64x64 sub-tile = 4x4 MFMA 16x16 accumulators (fp32).  The K step is
// 128 BYTES of the contraction dimension (64 bf16 / 32 fp32), so that the LDS image, the
// 16-byte fragment reads and the swizzle are identical for both storage types; only the
// MFMA differs (v_mfma_f32_16x16x32_bf16 vs four v_mfma_f32_16x16x4_f32).
//
// Orientation: the MFMA "A" operand (rows of D) is fed from W / dY^T and the "B" operand
// (columns of D) from the activation matrix, i.e. each wave computes a tile of C^T.  A
// lane then holds consecutive n for one m, which gives 16-element contiguous stores and
// a vectorised epilogue (bias/gamma as float4).
#include <stdlib.h>

#include <atomic>

#include "gemm_common.hpp"

namespace {

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmP p) {
    constexpr int EPV = TT<T>::EPV;
    constexpr int BK = 8 * EPV;  // elements per K tile (128 bytes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 buf][A, W][TILE_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int s = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * TILE, n0 = tn * TILE;

    // ---- global -> register staging: 4 x 16B chunks per operand per thread ----
    int ld_row[4], ld_ch[4];
    int64_t a_base[4];
    int64_t w_base[4];
    bool a_ok[4], w_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        ld_row[i] = q >> 3;
        ld_ch[i] = q & 7;
        const int m = m0 + ld_row[i];
        const int n = n0 + ld_row[i];
        a_ok[i] = m < p.M;
        w_ok[i] = n < p.N;
        const int mc = a_ok[i] ? m : 0;
        a_base[i] = p.a_mode == LNX_ADDR_PATCH2 ? patch_base(p.pg, mc) : (int64_t)mc * p.lda;
        w_base[i] = (int64_t)(w_ok[i] ? n : 0) * p.ldw;
    }
    uint4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + ld_ch[i] * EPV;
            const bool kin = kk < p.K;
            uint4 va = make_uint4(0, 0, 0, 0), vw = make_uint4(0, 0, 0, 0);
            if (a_ok[i] && kin) {
                const int64_t off = a_base[i] + (p.a_mode == LNX_ADDR_PATCH2 ? patch_col(p.pg, kk) : (int64_t)kk);
                va = ld16(p.A + off * sizeof(T));
            }
            if (w_ok[i] && kin) vw = ld16(p.W + (w_base[i] + kk) * sizeof(T));
            ra[i] = va;
            rw[i] = vw;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sa = smem + buf * 2 * TILE_BYTES;
        unsigned char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = ld_row[i] * ROWB + ((ld_ch[i] ^ row_key(ld_row[i])) << 4);
            st16(sa + off, ra[i]);
            st16(sw + off, rw[i]);
        }
    };

    // ---- fragment addressing (lane-constant) ----
    const int frag_row = (s >> 2) * 16 + (s & 3);           // + 4*i for fragment i
    const int frag_key = ((s >> 1) & 1) | ((s >> 2) << 1);  // == row_key(frag_row + 4*i + 64*w)
    const int a_row0 = wm * 64 + frag_row;
    const int w_row0 = wn * 64 + frag_row;

    f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* sa = smem + buf * 2 * TILE_BYTES;
        const unsigned char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((g + 4 * kk) ^ frag_key) << 4;
            uint4 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = ld16(sw + (w_row0 + 4 * i) * ROWB + ch);
                af[i] = ld16(sa + (a_row0 + 4 * i) * ROWB + ch);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc[ni][mi], wf[ni], af[mi]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<T, OUT_F32>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------
// weight-gradient GEMM (contraction over the row index M of both operands)
// ------------------------------------------------------------------------------------

template <typename T> struct WgradCfg;
template <> struct WgradCfg<bf16_t> {
    static constexpr int BMC = 64;              // rows of M per LDS tile
    static constexpr int ROW = 128 * 2 + 32;    // bytes per LDS row (+32B pad: tr reads conflict-free)
};
template <> struct WgradCfg<float> {
    static constexpr int BMC = 32;
    static constexpr int ROW = 128 * 4 + 64;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ uint2 ds_read_tr16_b64(const unsigned char* p) {
    // 4 rows x 16 columns of 16-bit elements per 16-lane group, delivered column-major.
    // EXEC must be all ones (the gather crosses lanes): only called from uniform code.
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    return __builtin_bit_cast(uint2, v);
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const WgradP p) {
    constexpr int EPV = TT<T>::EPV;
    constexpr int BMC = WgradCfg<T>::BMC;
    constexpr int ROW = WgradCfg<T>::ROW;
    constexpr int OPB = BMC * ROW;             // bytes per operand tile
    constexpr int CPR = 128 / EPV;             // 16-byte chunks per tile row
    constexpr int NLD = BMC * CPR / 256;       // chunks per thread per operand
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 buf][dY, A][OPB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int s = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    const int split = bid % p.splits;
    bid /= p.splits;
    const int tk = bid % p.tiles_k;
    const int tn = bid / p.tiles_k;
    const int n0 = tn * TILE, k0 = tk * TILE;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;

    int ld_row[NLD], ld_ch[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int q = tid + 256 * i;
        ld_row[i] = q / CPR;
        ld_ch[i] = q % CPR;
    }
    uint4 ry[NLD], ra[NLD];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = mt + ld_row[i];
            const int n = n0 + ld_ch[i] * EPV;
            const int k = k0 + ld_ch[i] * EPV;
            uint4 vy = make_uint4(0, 0, 0, 0), va = make_uint4(0, 0, 0, 0);
            if (m < m_end) {
                if (n < p.N) vy = ld16(p.dY + ((int64_t)m * p.lddy + n) * sizeof(T));
                if (k < p.K) {
                    const int64_t off = p.a_mode == LNX_ADDR_PATCH2 ? patch_base(p.pg, m) + patch_col(p.pg, k) : (int64_t)m * p.lda + k;
                    va = ld16(p.A + off * sizeof(T));
                }
            }
            ry[i] = vy;
            ra[i] = va;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* sy = smem + buf * 2 * OPB;
        unsigned char* sa = sy + OPB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int off = ld_row[i] * ROW + ld_ch[i] * 16;
            st16(sy + off, ry[i]);
            st16(sa + off, ra[i]);
        }
    };

    f32x4_t acc[4][4];  // [ni][ki]
    f32x4_t accb[4];    // bias: colsum(dY) via MFMA against a ones operand
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.db != nullptr && tk == 0 && wk == 0;

    const int ntile = (m_end - m_begin + BMC - 1) / BMC;
    load_tile(m_begin);
    store_tile(0);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        const int buf = it & 1;
        if (it + 1 < ntile) load_tile(m_begin + (it + 1) * BMC);
        const unsigned char* sy = smem + buf * 2 * OPB;
        const unsigned char* sa = sy + OPB;
        if constexpr (sizeof(T) == 2) {
            // contraction slot (g, j) of a 32-row step <-> row 4g + j (j<4), 16 + 4g + (j-4)
            const int q = s >> 2, pp = s & 3;
#pragma unroll
            for (int ks = 0; ks < BMC / 32; ++ks) {
                uint4 yf[4], af[4];
                const int r0 = ks * 32 + 4 * g + q;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cy = (wn * 64 + i * 16 + 4 * pp) * 2;
                    const int ca = (wk * 64 + i * 16 + 4 * pp) * 2;
                    const uint2 y0 = ds_read_tr16_b64(sy + r0 * ROW + cy);
                    const uint2 y1 = ds_read_tr16_b64(sy + (r0 + 16) * ROW + cy);
                    const uint2 a0 = ds_read_tr16_b64(sa + r0 * ROW + ca);
                    const uint2 a1 = ds_read_tr16_b64(sa + (r0 + 16) * ROW + ca);
                    yf[i] = make_uint4(y0.x, y0.y, y1.x, y1.y);
                    af[i] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int ki = 0; ki < 4; ++ki) Mfma<T>::run(acc[ni][ki], yf[ni], af[ki]);
                if (do_bias) {
                    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) Mfma<T>::run(accb[ni], yf[ni], ones);
                }
            }
        } else {
#pragma unroll
            for (int ms = 0; ms < BMC / 4; ++ms) {
                float yf[4], af[4];
                const int r = ms * 4 + g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    yf[i] = *reinterpret_cast<const float*>(sy + r * ROW + (wn * 64 + i * 16 + s) * 4);
                    af[i] = *reinterpret_cast<const float*>(sa + r * ROW + (wk * 64 + i * 16 + s) * 4);
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int ki = 0; ki < 4; ++ki)
                        acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[ni], af[ki], acc[ni][ki], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) accb[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(yf[ni], 1.0f, accb[ni], 0, 0, 0);
                }
            }
        }
        if (it + 1 < ntile) store_tile(buf ^ 1);
        __syncthreads();
    }

    // D[row = n slot][col = k slot]: lane (s, g) holds n = .. + 4g + r, k = .. + s
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * 64 + ni * 16 + 4 * g + r;
            if (n >= p.N) continue;
#pragma unroll
            for (int ki = 0; ki < 4; ++ki) {
                const int k = k0 + wk * 64 + ki * 16 + s;
                if (k >= p.k_store) continue;
                int col = k;
                if (p.k_perm_c > 0) {
                    const int P = p.K / p.k_perm_c;
                    const int pp = k / p.k_perm_c;
                    col = (k - pp * p.k_perm_c) * P + pp;
                }
                atomicAdd(p.dW + (int64_t)n * p.lddw + col, acc[ni][ki][r]);
            }
            if (do_bias && s == 0) atomicAdd(p.db + n, accb[ni][r]);
        }
    }
}

template <typename T, bool OUT_F32>
int launch_nt(const GemmP& p, hipStream_t st) {
    const int grid = p.tiles_m * p.tiles_n;
    const size_t lds = 4 * TILE_BYTES;
    hipLaunchKernelGGL((gemm_nt_kernel<T, OUT_F32>), dim3(grid), dim3(256), lds, st, p);
    return 0;
}

template <typename T>
int launch_tn(const WgradP& p, hipStream_t st) {
    const int grid = p.tiles_n * p.tiles_k * p.splits;
    const size_t lds = 4 * WgradCfg<T>::BMC * WgradCfg<T>::ROW;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(grid), dim3(256), lds, st, p);
    return 0;
}

}  // namespace

static bool g_force_v1 = getenv("LNX_GEMM_V1") != nullptr;  // A/B switch for benchmarking

// dispatch bookkeeping (include/lnx.h: lnx_last_nt_kernel / lnx_nt_kernel_launches)
static std::atomic<int> g_last_nt{0};
static std::atomic<long long> g_nt_launches[LNX_NT_KERNEL_KINDS];
namespace lnxg {
void note_nt_kernel(int kind) {
    if (kind < 0 || kind >= LNX_NT_KERNEL_KINDS) return;
    g_last_nt.store(kind, std::memory_order_relaxed);
    g_nt_launches[kind].fetch_add(1, std::memory_order_relaxed);
}
}  // namespace lnxg
extern "C" int lnx_last_nt_kernel(void) { return g_last_nt.load(std::memory_order_relaxed); }
extern "C" int64_t lnx_nt_kernel_launches(int kind) {
    return (kind < 0 || kind >= LNX_NT_KERNEL_KINDS) ? -1 : (int64_t)g_nt_launches[kind].load(std::memory_order_relaxed);
}

static void fill_gemm_p(const lnx_gemm_args* a, GemmP& p) {
    p.A = (const unsigned char*)a->A;
    p.W = (const unsigned char*)a->W;
    p.C = (unsigned char*)a->C;
    p.C2 = (unsigned char*)a->c2;
    p.aux = (const unsigned char*)a->aux;
    p.bias = a->bias;
    p.gamma = a->gamma;
    p.rowscale = a->rowscale;
    p.res = a->res;
    p.lda = a->lda;
    p.ldw = a->ldw;
    p.ldc = a->ldc;
    p.ldc2 = a->ldc2;
    p.ldaux = a->ldaux;
    p.ldres = a->ldres;
    p.M = a->M;
    p.N = a->N;
    p.K = a->K;
    p.a_mode = a->a_mode;
    p.c_mode = a->c_mode;
    p.pg = PatchGeom{a->Hin, a->Win, a->Cin};
    p.cmap = RowMap{a->c_map.group, a->c_map.pad, a->c_map.off};
    p.act = a->act;
    p.rows_per_sample = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
    p.tiles_m = cdiv(a->M, TILE);
    p.tiles_n = cdiv(a->N, TILE);
}

// The dispatcher's decision without a launch (include/lnx.h): only M / N / K / dtype / out_f32 / act / addressing modes and WHICH of the
// optional operands are present matter -- the pointers are never dereferenced, so a host-side caller may pass any non-null value.
extern "C" int lnx_nt_dispatch(const lnx_gemm_args* a) {
    if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->dtype != LNX_F32 && a->dtype != LNX_BF16)) return -1;
    GemmP p;
    fill_gemm_p(a, p);
    if (nt_skinny_ok(p, a->dtype, a->out_f32 != 0) && !g_force_v1) return LNX_NT_KERNEL_SKINNY;
    if (nt_v2_ok(p, a->dtype) && !g_force_v1) return nt_v2_family(p, a->out_f32 != 0, nullptr);
    return LNX_NT_KERNEL_V1;
}

extern "C" int lnx_gemm_nt(const lnx_gemm_args* a, void* stream) {
    LNX_CHECK(a != nullptr, "lnx_gemm_nt: null args");
    LNX_CHECK(a->dtype == LNX_F32 || a->dtype == LNX_BF16, "lnx_gemm_nt: bad dtype %d", a->dtype);
    const int epv = a->dtype == LNX_F32 ? 4 : 8;
    LNX_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "lnx_gemm_nt: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    LNX_CHECK(a->K % epv == 0, "lnx_gemm_nt: K=%d must be a multiple of %d", a->K, epv);
    LNX_CHECK(a->A && a->W && a->C, "lnx_gemm_nt: null operand");
    LNX_CHECK(a->ldw % epv == 0, "lnx_gemm_nt: ldw=%lld must be a multiple of %d", (long long)a->ldw, epv);
    LNX_CHECK((((uintptr_t)a->A) & 15) == 0 && (((uintptr_t)a->W) & 15) == 0, "lnx_gemm_nt: A/W must be 16-byte aligned");
    if (a->a_mode == LNX_ADDR_PATCH2) {
        LNX_CHECK(a->Hin % 2 == 0 && a->Win % 2 == 0 && a->Cin % epv == 0, "lnx_gemm_nt: bad PATCH2 geometry %dx%dx%d", a->Hin, a->Win, a->Cin);
        LNX_CHECK(a->K == 4 * a->Cin, "lnx_gemm_nt: PATCH2 needs K == 4*Cin");
        LNX_CHECK(a->M % ((a->Hin / 2) * (a->Win / 2)) == 0, "lnx_gemm_nt: PATCH2 needs M == B*Ho*Wo");
    } else {
        LNX_CHECK(a->lda % epv == 0, "lnx_gemm_nt: lda=%lld must be a multiple of %d", (long long)a->lda, epv);
    }
    if (a->c_mode == LNX_ADDR_PATCH2) {
        LNX_CHECK(a->Hin % 2 == 0 && a->Win % 2 == 0 && a->Cin % 16 == 0, "lnx_gemm_nt: bad PATCH2 output geometry");
        LNX_CHECK(a->N == 4 * a->Cin, "lnx_gemm_nt: PATCH2 output needs N == 4*Cin");
        LNX_CHECK(a->a_mode == LNX_ADDR_PLAIN, "lnx_gemm_nt: PATCH2 on both sides is not supported");
    }
    LNX_CHECK(a->act >= LNX_ACT_NONE && a->act <= LNX_ACT_MUL_AUX, "lnx_gemm_nt: unknown act %d", a->act);
    if (a->act == LNX_ACT_GELU_BWD || a->act == LNX_ACT_RELU_BWD || a->act == LNX_ACT_MUL_AUX) LNX_CHECK(a->aux != nullptr, "lnx_gemm_nt: act %d needs aux", a->act);
    if (a->act == LNX_ACT_GELU_D) LNX_CHECK(a->c2 != nullptr, "lnx_gemm_nt: GELU_D writes the derivative to c2, which is NULL");
    if (a->rowscale) LNX_CHECK(a->rows_per_sample > 0, "lnx_gemm_nt: rowscale needs rows_per_sample");

    GemmP p;
    fill_gemm_p(a, p);
    hipStream_t st = (hipStream_t)stream;
    if (nt_skinny_ok(p, a->dtype, a->out_f32 != 0) && !g_force_v1) {
        note_nt_kernel(LNX_NT_KERNEL_SKINNY);
        launch_nt_skinny(p, a->out_f32 != 0, st);
    } else if (nt_v2_ok(p, a->dtype) && !g_force_v1) {
        launch_nt_v2(p, a->out_f32 != 0, st);  // (notes its own choice: v2 / v4 / v7 / v9)
    } else if (a->dtype == LNX_BF16) {
        note_nt_kernel(LNX_NT_KERNEL_V1);
        if (a->out_f32) launch_nt<bf16_t, true>(p, st);
        else launch_nt<bf16_t, false>(p, st);
    } else {
        note_nt_kernel(LNX_NT_KERNEL_V1);
        launch_nt<float, true>(p, st);  // T == float: both output kinds are fp32
    }
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_gemm_nt_group_ok(const lnx_gemm_args* a, int n, int accumulate) {
    if (!a || n < 1 || n > LNX_GEMM_GROUP_MAX) return 0;
    for (int j = 0; j < n; ++j) {
        const lnx_gemm_args& t = a[j];
        if (t.dtype != LNX_BF16 || t.M <= 0 || t.N <= 0 || t.K <= 0 || t.K % 8 != 0 || !t.A || !t.W || t.lda % 8 != 0 || t.ldw % 8 != 0) return 0;
        if ((((uintptr_t)t.A) & 15) != 0 || (((uintptr_t)t.W) & 15) != 0) return 0;
        if (t.a_mode != LNX_ADDR_PLAIN || t.c_mode != LNX_ADDR_PLAIN || t.c_map.group > 0) return 0;
        if (t.act != LNX_ACT_NONE || t.gamma || t.rowscale || t.aux || t.c2 || t.c8) return 0;
        if (t.out_f32 != a[0].out_f32) return 0;
        if (accumulate && j > 0) {
            if (t.M != a[0].M || t.N != a[0].N || t.bias || t.res) return 0;  // (term 0 carries the output and its epilogue operands)
        } else {
            if (!t.C || (t.res && !t.out_f32) || (t.out_f32 && (((uintptr_t)t.C) & 15) != 0)) return 0;
        }
    }
    return 1;
}

extern "C" int lnx_gemm_nt_group(const lnx_gemm_args* a, int n, int accumulate, void* stream) {
    LNX_CHECK(lnx_gemm_nt_group_ok(a, n, accumulate), "lnx_gemm_nt_group: 1..%d bf16 problems with plain addressing, bias / fp32 residual epilogues only, K and "
              "leading dimensions multiples of 8, 16-byte aligned operands; accumulate: same M, N and the output / bias / res in term 0", LNX_GEMM_GROUP_MAX);
    GemmP ps[LNX_GEMM_GROUP_MAX];
    for (int j = 0; j < n; ++j) fill_gemm_p(&a[j], ps[j]);
    note_nt_kernel(LNX_NT_KERNEL_SKINNY);
    launch_nt_skinny_group(ps, n, accumulate != 0, a[0].out_f32 != 0, (hipStream_t)stream);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_gemm_tn(const lnx_wgrad_args* a, void* stream) {
    LNX_CHECK(a != nullptr, "lnx_gemm_tn: null args");
    LNX_CHECK(a->dtype == LNX_F32 || a->dtype == LNX_BF16, "lnx_gemm_tn: bad dtype %d", a->dtype);
    const int epv = a->dtype == LNX_F32 ? 4 : 8;
    LNX_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "lnx_gemm_tn: empty problem");
    LNX_CHECK(a->dY && a->A && a->dW, "lnx_gemm_tn: null operand");
    // dY rows are read in 16-byte chunks: a chunk that starts below N may run past it, so the row
    // must be padded (lddy >= roundup(N, epv)); what the padding holds is never stored
    LNX_CHECK(a->lddy % epv == 0 && a->lddy >= (int64_t)cdiv(a->N, epv) * epv, "lnx_gemm_tn: lddy=%lld must be a multiple of %d and >= roundup(N)", (long long)a->lddy, epv);
    LNX_CHECK(a->K % epv == 0, "lnx_gemm_tn: K=%d must be a multiple of %d", a->K, epv);
    LNX_CHECK((((uintptr_t)a->A) & 15) == 0 && (((uintptr_t)a->dY) & 15) == 0, "lnx_gemm_tn: operands must be 16-byte aligned");
    if (a->a_mode == LNX_ADDR_PATCH2) {
        LNX_CHECK(a->Hin % 2 == 0 && a->Win % 2 == 0 && a->Cin % epv == 0 && a->K == 4 * a->Cin, "lnx_gemm_tn: bad PATCH2 geometry");
    } else {
        LNX_CHECK(a->lda % epv == 0, "lnx_gemm_tn: lda must be a multiple of %d", epv);
    }
    if (a->k_perm_c > 0) LNX_CHECK(a->K % a->k_perm_c == 0, "lnx_gemm_tn: K %% k_perm_c != 0");
    WgradP p;
    p.dY = (const unsigned char*)a->dY;
    p.A = (const unsigned char*)a->A;
    p.dW = a->dW;
    p.db = a->db;
    p.lddy = a->lddy;
    p.lda = a->lda;
    p.lddw = a->lddw;
    p.M = a->M;
    p.N = a->N;
    p.K = a->K;
    p.a_mode = a->a_mode;
    p.pg = PatchGeom{a->Hin, a->Win, a->Cin};
    p.k_perm_c = a->k_perm_c;
    p.ws = a->ws;
    p.ws_floats = a->ws ? a->ws_floats : 0;
    p.k_store = (a->k_store > 0 && a->k_store < a->K) ? a->k_store : a->K;
    p.tiles_n = cdiv(a->N, TILE);
    p.tiles_k = cdiv(a->K, TILE);
    const int bmc = a->dtype == LNX_BF16 ? 64 : 32;
    int splits = a->splits;
    if (splits <= 0) {
        // aim at ~2 workgroups per CU, but keep >= 4 M-tiles of work per split
        const int tiles = p.tiles_n * p.tiles_k;
        splits = cdiv(2 * 256, tiles);
        const int max_splits = cdiv(a->M, 4 * bmc);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int mps = cdiv(a->M, splits);
    mps = cdiv(mps, bmc) * bmc;
    p.splits = cdiv(a->M, mps);
    p.m_per_split = mps;
    hipStream_t st = (hipStream_t)stream;
    static const bool no_defer = getenv("LNX_TN_NO_DEFER") != nullptr;  // A/B switch: every product reduces its own partial tiles at once
    if (tn_v2_ok(p, a->dtype) && !g_force_v1) launch_tn_v2(p, a->splits, st, a->defer != 0 && !no_defer);
    else if (a->dtype == LNX_BF16) launch_tn<bf16_t>(p, st);
    else launch_tn<float>(p, st);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_gemm_tn_flush(void* stream) {
    LNX_CHECK(tn_flush((hipStream_t)stream) == 0, "lnx_gemm_tn_flush: the postponed products of this thread were launched on another stream (flush on that stream, or lnx_gemm_tn_discard)");
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_gemm_tn_discard(void) { return tn_discard(); }
