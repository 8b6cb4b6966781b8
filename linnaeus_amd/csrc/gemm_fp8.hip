// fp8 (OCP e4m3fn) operands for the NT GEMM on gfx950: lnx_amax / lnx_quantize_fp8 / lnx_gemm_nt_fp8 (include/lnx.h).
//
// The GEMM is the 256x128-tile LDS-DMA pipeline of gemm2.hip with the same bytes per stage: a 128-byte LDS row is one
// K slice of 128 e4m3 values instead of 64 bf16, so each K slice does twice the work for the same fill traffic, and
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; the per-tensor scales are applied to the accumulator) retires
// it at twice the bf16 MFMA rate.  Operand lane map, verified with exact integer data (tools/ubench/mfma_f8_layout.hip):
// lane l holds A[row l & 15][k = 32 (l >> 4) + j] in byte j of its 8 dwords, B likewise with the column on l & 15; C/D as
// every 16x16 MFMA.  A lane's 32 bytes are the two 16-byte chunks 2g, 2g+1 of the LDS row (XOR swizzle per chunk).
//
// MXFP8 (lnx_quantize_mxfp8 / lnx_gemm_nt_mxfp8): the same pipeline with the instruction's block scales in use -- one
// E8M0 (power-of-two) scale per 32 consecutive K elements of a row, applied by the matrix core itself.  What the hardware
// takes as "k" (tools/ubench/mfma_mx_map.hip, mfma_mx_scale.hip): a lane's dwords 0-3 are k = 16 g .. 16 g + 15, its
// dwords 4-7 are k = 64 + 16 g .. 64 + 16 g + 15, and the scale of block kb of row r is the selected byte of lane
// 16 kb + r's scale register.  So the MX kernel reads chunks g and g + 4 of the LDS row, and its scales travel with the
// tile: one more LDS-DMA instruction per wave and stage brings a dword (the four block scales of this K slice) per tile
// row from the [K/128][rows] scale array, and each lane picks its byte with ds_read_u8.
#include "gemm_common.hpp"

namespace lnxg {

#define F8_DS_READ4(dst, addr)                                                                           \
    do {                                                                                                 \
        const uint32_t a_ = (addr);                                                                      \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                         \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[1]) : "v"(a_) : "memory");              \
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:1536" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

constexpr int F8_BM = 256, F8_BN = 128, F8_BK = 128;   // BK in elements = bytes
constexpr int F8_STAGE = (F8_BM + F8_BN) * ROWB;       // 48 KiB
constexpr int F8_SCALES = 8 * 64 * 4;                  // MX: one dword per tile row (384 used), one 256-byte DMA per wave
constexpr int F8_NSTAGE = 3;
constexpr int F8_LD = (F8_BM + F8_BN) / 8 / 8;         // 1-KiB LDS-DMA instructions per wave per stage = 6

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

__device__ __forceinline__ void mfma_f8(f32x4_t& acc, const uint4& a_lo, const uint4& a_hi, const uint4& b_lo, const uint4& b_hi, int sa = 0x7f, int sb = 0x7f) {
    const i32x8_t a = {(int)a_lo.x, (int)a_lo.y, (int)a_lo.z, (int)a_lo.w, (int)a_hi.x, (int)a_hi.y, (int)a_hi.z, (int)a_hi.w};
    const i32x8_t b = {(int)b_lo.x, (int)b_lo.y, (int)b_lo.z, (int)b_lo.w, (int)b_hi.x, (int)b_hi.y, (int)b_hi.z, (int)b_hi.w};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);  // e4m3 x e4m3; scale = byte 0 of sa / sb (E8M0, 0x7f = 2^0)
}

template <bool OUT_F32, int F, bool MX>
__global__ __launch_bounds__(512) void gemm_nt_fp8_kernel(const GemmP p) {
    constexpr int STAGE = F8_STAGE + (MX ? F8_SCALES : 0);
    constexpr int NLD = F8_LD + (MX ? 1 : 0);  // VMEM instructions per wave and stage
    typedef bf16_t T;  // type of C / c2 / aux
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3][A 256 rows | W 128 rows][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int s = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * F8_BM, n0 = tn * F8_BN;

    // LDS-DMA sources as in gemm_nt_v2_kernel (bytes == elements)
    const unsigned char* src[F8_LD];
#pragma unroll
    for (int j = 0; j < F8_LD; ++j) {
        const int i = wave + 8 * j;
        const int row = 8 * i + (lane >> 3);
        const int slot = lane & 7;
        if (row < F8_BM) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[j] = p.A + (int64_t)m * p.lda + ((slot ^ row_key(row)) << 4);
        } else {
            const int wr = row - F8_BM;
            int n = n0 + wr;
            if (n >= p.N) n = p.N - 1;
            src[j] = p.W + (int64_t)n * p.ldw + ((slot ^ row_key(wr)) << 4);
        }
    }
    // MX: lane `lane` of wave `wave` fetches the scale dword of tile row 64 wave + lane (A rows 0..255, W rows 256..383,
    // the last two waves re-fetch the last W row: every wave issues the same number of VMEM instructions)
    const uint32_t* sc_src = nullptr;
    int64_t sc_step = 0;
    if constexpr (MX) {
        const int row = 64 * wave + lane;
        if (row < F8_BM) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            sc_src = p.mxa + m;
            sc_step = p.M;
        } else {
            int n = n0 + row - F8_BM;
            if (row >= F8_BM + F8_BN || n >= p.N) n = p.N - 1;
            sc_src = p.mxw + n;
            sc_step = p.N;
        }
    }
    auto issue_tile = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < F8_LD; ++j) {
            const int i = wave + 8 * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (int64_t)kt * F8_BK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + i * 1024), 16, 0, 0);
        }
        if constexpr (MX)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc_src + (int64_t)kt * sc_step),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + F8_STAGE + wave * 256), 4, 0, 0);
    };

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const int frag_key = ((s >> 1) & 1) | ((s >> 2) << 1);
    const int a_row0 = wm * 64 + frag_row;
    const int w_row0 = F8_BM + wn * 64 + frag_row;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // plain: bytes 32g .. 32g+31 of the row (any k order does, as long as A and W agree); MX: the hardware's own k order
    const uint32_t ch0 = (uint32_t)(((MX ? g : 2 * g) ^ frag_key) << 4), ch1 = (uint32_t)(((MX ? g + 4 : 2 * g + 1) ^ frag_key) << 4);
    const uint32_t sc_a0 = F8_STAGE + (wm * 64 + frag_row) * 4 + g, sc_w0 = F8_STAGE + (F8_BM + wn * 64 + frag_row) * 4 + g;
    const uint32_t a_off0 = a_row0 * ROWB + ch0, a_off1 = a_row0 * ROWB + ch1;
    const uint32_t w_off0 = w_row0 * ROWB + ch0, w_off1 = w_row0 * ROWB + ch1;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / F8_BK;  // >= 2
    issue_tile(0, 0);
    issue_tile(1, 1);
    // ping-pong wave groups, two barriers per K slice, counted waits: see gemm_nt_v2_kernel
    const int grp = wave >> 2;
    if constexpr (MX) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t st = lds_base + (kt % F8_NSTAGE) * STAGE;
        uint4 wf0[4], af0[4], wf1[4], af1[4];
        F8_DS_READ4(wf0, st + w_off0);
        F8_DS_READ4(af0, st + a_off0);
        F8_DS_READ4(wf1, st + w_off1);
        F8_DS_READ4(af1, st + a_off1);
        int scw[4], sca[4];
        if constexpr (MX) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // fragment i of this lane is tile row frag_row + 4 i: 16 bytes further on
                asm volatile("ds_read_u8 %0, %1" : "=v"(scw[i]) : "v"(st + sc_w0 + 16 * i) : "memory");
                asm volatile("ds_read_u8 %0, %1" : "=v"(sca[i]) : "v"(st + sc_a0 + 16 * i) : "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) scw[i] = sca[i] = 0x7f;
        }
        if (kt + 2 < nk) {
            issue_tile(kt + 2, (kt + 2) % F8_NSTAGE);
            if constexpr (MX) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mfma_f8(acc[ni][mi], wf0[ni], wf1[ni], af0[mi], af1[mi], scw[ni], sca[mi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp && kt + 1 == nk)) __builtin_amdgcn_s_barrier();
    }
    if constexpr (!MX) {
        const float alpha = (p.sa ? p.sa[0] : 1.0f) * (p.sw ? p.sw[0] : 1.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= alpha;
    }
    gemm_epilogue_fast<T, OUT_F32, F>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// MXFP8 on the 256x256 tile (round 3): the byte flow of gemm_nt_v4 (32 KiB stages of 64-BYTE rows, four of them, 8 waves with
// 128x64 wave tiles in two ping-pong groups) with v_mfma_scale_f32_32x32x64_f8f6f4 -- a K slice is 64 e4m3 values = two MX blocks
// per row, so a stage carries twice the work of a bf16 stage of the same size: the fill traffic per FLOP of the 256x128 fp8
// kernel above is halved again.  Operand / scale layout of the instruction, checked with exact data
// (tools/ubench/mfma_mx32_layout.hip): lane l = (r = l & 31, h = l >> 5) supplies row r; its dwords 0-3 belong to MX block 0 of
// the slice, dwords 4-7 to block 1 (h picks the 16-byte half of the block: any order inside a block does, as long as A and W
// agree); the scale of block kb of row r is byte 0 of lane (r + 32 kb)'s scale register; result register i of lane l is
// D[8 (i / 4) + 4 h + i % 4][r].  The W rows of a 32-row block are loaded PERMUTED (MFMA row j <- block row
// 16 ((j >> 2) & 1) + 4 (j >> 3) + (j & 3)) so that a lane's 16 results are 16 consecutive output columns.
// ---------------------------------------------------------------------------------------------------------------
constexpr int X8_BM = 256, X8_BN = 256, X8_BK = 64, X8_ROWB = 64;
constexpr int X8_STAGE = (X8_BM + X8_BN) * X8_ROWB;  // 32 KiB
constexpr int X8_SCALES = 512 * 4;                    // one dword (the four block scales of a 128-wide K span) per tile row
constexpr int X8_NST = 4;
constexpr int X8_PIECES = X8_STAGE / 1024 / 8;        // 4 LDS-DMA wave-instructions per wave and stage (+ 1 for the scales)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// chunk key of a 64-byte LDS row: conflict-free for 16 consecutive rows (the A fragments) and for the permuted W rows
__device__ __forceinline__ int key_x8(int row) { return ((row >> 2) & 3) ^ (((row >> 4) & 1) << 1); }

template <bool OUT_F32, int F>
__device__ __forceinline__ void mx8_epilogue(const GemmP& p, f32x16_t (&acc)[2][4], int mrow0, int ncol0, int lane) {
    typedef bf16_t T;
    const int r = lane & 31, h = lane >> 5;
    float bias[2][16];
    if (F & F_BIAS) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(p.bias + ncol0 + 32 * nj + 16 * h + 4 * q);
                bias[nj][4 * q] = t.x; bias[nj][4 * q + 1] = t.y; bias[nj][4 * q + 2] = t.z; bias[nj][4 * q + 3] = t.w;
            }
    }
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {  // two row blocks at a time: their operand fetches go out together
        float ld[2][2][16];
        float rs[2] = {1.f, 1.f};
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
            const int m = min(mrow0 + 32 * (2 * mp + mq) + r, p.M - 1);
            if (F & F_RES) {
                if (p.rowscale) rs[mq] = p.rowscale[m / p.rows_per_sample];
            }
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int nb = ncol0 + 32 * nj + 16 * h;
                if (F & F_GELU_BWD) {
                    const T* ax = reinterpret_cast<const T*>(p.aux) + ((int64_t)m * p.ldaux + nb);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        Vec16<T> t;
                        t.raw = ld16(ax + 8 * hh);
#pragma unroll
                        for (int j = 0; j < 8; ++j) ld[mq][nj][8 * hh + j] = t.get(j);
                    }
                }
                if (F & F_RES) {
                    const float* rp = p.res + ((int64_t)m * p.ldres + nb);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(rp + 4 * q);
                        ld[mq][nj][4 * q] = t.x; ld[mq][nj][4 * q + 1] = t.y; ld[mq][nj][4 * q + 2] = t.z; ld[mq][nj][4 * q + 3] = t.w;
                    }
                }
            }
        }
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
            const int mj = 2 * mp + mq;
            const int m = mrow0 + 32 * mj + r;
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int nb = ncol0 + 32 * nj + 16 * h;
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (F & F_BIAS) ? acc[nj][mj][i] + bias[nj][i] : acc[nj][mj][i];
                const bool live = m < p.M;  // (the exchange of the MX output below needs every lane: no early exit)
                if (F & F_C2) {
                    if (live) {
                        T* c2 = reinterpret_cast<T*>(p.C2) + ((int64_t)m * p.ldc2 + nb);
                        Vec16<T> o;
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) o.set(j, v[8 * hh + j]);
                            st16(c2 + 8 * hh, o.raw);
                        }
                    }
                }
                if (F & F_GELU) Gelu<T>::fwd16(v);
                if (F & F_GELU_BWD) Gelu<T>::mulgrad16(v, ld[mq][nj]);
                if (F & F_RES) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], rs[mq], ld[mq][nj][j]);
                }
                if (OUT_F32) {
                    if (live) {
                        float* cp = reinterpret_cast<float*>(p.C) + ((int64_t)m * p.ldc + nb);
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(cp + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    }
                } else {
                    T* cp = reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + nb);
                    Vec16<T> o;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o.set(j, v[8 * hh + j]);
                        if (live) st16(cp + 8 * hh, o.raw);
                        if (F & F_MXOUT) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[8 * hh + j] = o.get(j);  // the stored (rounded) values are what gets quantised
                        }
                    }
                    if (F & F_MXOUT) {
                        // a 32-element block = this lane's 16 columns and those of lane ^ 32 (same row, the other half)
                        float am = 0.f;
#pragma unroll
                        for (int j = 0; j < 16; ++j) am = fmaxf(am, fabsf(v[j]));
                        am = fmaxf(am, __shfl_xor(am, 32, 64));
                        const uint32_t bits = __float_as_uint(am);
                        int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
                        e = e < 0 ? 0 : (e > 254 ? 254 : e);
                        const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
                        uint32_t w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * q] * inv, -448.f), 448.f), fminf(fmaxf(v[4 * q + 1] * inv, -448.f), 448.f), 0, false);
                            pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[4 * q + 2] * inv, -448.f), 448.f), fminf(fmaxf(v[4 * q + 3] * inv, -448.f), 448.f), pk, true);
                            w[q] = (uint32_t)pk;
                        }
                        if (live) {
                            st16(p.C8 + (int64_t)m * p.ldc8 + nb, make_uint4(w[0], w[1], w[2], w[3]));
                            const int kb = nb >> 5;
                            if (h == 0) p.C8s[((int64_t)(kb >> 2) * p.M + m) * 4 + (kb & 3)] = (unsigned char)e;
                        }
                    }
                }
            }
        }
    }
}

template <bool OUT_F32, int F>
__global__ __launch_bounds__(512) void gemm_nt_mx8_kernel(const GemmP p) {
    constexpr int STAGE = X8_STAGE + X8_SCALES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [4][A 256 rows | W 256 rows][64 B] + [512] scale dwords
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // 128-row half, 64-column quarter; waves 0-3 / 4-7 = the ping-pong groups
    const int r = lane & 31, h = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * X8_BM, n0 = tn * X8_BN;

    // piece i (32 per stage) fills LDS rows 16 i .. 16 i + 15 (rows 0..255 = A, 256..511 = W); wave w issues i = w + 8 j
    const unsigned char* src[X8_PIECES];
#pragma unroll
    for (int j = 0; j < X8_PIECES; ++j) {
        const int i = wave + 8 * j;
        const int row = 16 * i + (lane >> 2);
        const int slot = lane & 3;
        if (row < X8_BM) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[j] = p.A + (int64_t)m * p.lda + ((slot ^ key_x8(row)) << 4);
        } else {
            const int wr = row - X8_BM;
            int n = n0 + wr;
            if (n >= p.N) n = p.N - 1;
            src[j] = p.W + (int64_t)n * p.ldw + ((slot ^ key_x8(wr)) << 4);
        }
    }
    // scales: lane `lane` of wave `wave` fetches the dword of tile row 64 wave + lane (A rows 0..255, W rows 256..511): the four
    // block scales of the 128-wide K span that holds this slice (fetched with every slice: one more VMEM instruction per stage)
    const uint32_t* sc_src;
    int64_t sc_step;
    {
        const int row = 64 * wave + lane;
        if (row < X8_BM) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            sc_src = p.mxa + m;
            sc_step = p.M;
        } else {
            int n = n0 + row - X8_BM;
            if (n >= p.N) n = p.N - 1;
            sc_src = p.mxw + n;
            sc_step = p.N;
        }
    }
    auto issue_stage = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < X8_PIECES; ++j) {
            const int i = wave + 8 * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (int64_t)kt * X8_BK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + i * 1024), 16, 0, 0);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc_src + (int64_t)(kt >> 1) * sc_step),
                                         (__attribute__((address_space(3))) void*)(smem + stage * STAGE + X8_STAGE + wave * 256), 4, 0, 0);
    };

    // fragment rows: A natural (block row r), W permuted (see above)
    const int wperm = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    uint32_t a_off[4][2], w_off[2][2], a_sc[4], w_sc[2];  // [block][MX block of the slice]
#pragma unroll
    for (int mj = 0; mj < 4; ++mj) {
        const int row = wm * 128 + 32 * mj + r;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) a_off[mj][kb] = (uint32_t)(row * X8_ROWB + (((2 * kb + h) ^ key_x8(row)) << 4));
        a_sc[mj] = (uint32_t)(X8_STAGE + row * 4 + h);
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int wr = wn * 64 + 32 * nj + wperm;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) w_off[nj][kb] = (uint32_t)((X8_BM + wr) * X8_ROWB + (((2 * kb + h) ^ key_x8(wr)) << 4));
        w_sc[nj] = (uint32_t)(X8_STAGE + (X8_BM + wr) * 4 + h);
    }

    f32x16_t acc[2][4];  // [nj][mj]
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mj = 0; mj < 4; ++mj)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nj][mj][i] = 0.f;

    const int nk = p.K / X8_BK;  // >= 4
    issue_stage(0, 0);
    issue_stage(1, 1);
    issue_stage(2, 2);
    const int grp = wave >> 2;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (X8_PIECES + 1)) : "memory");  // own pieces of slice 0 landed
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t st = lds_base + (kt % X8_NST) * STAGE;
        const uint32_t sb = (uint32_t)(2 * (kt & 1));  // which two of the dword's four block scales
        uint4 af[4][2], wf[2][2];
        int sca[4], scw[2];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(wf[nj][0]) : "v"(st + w_off[nj][0]) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(wf[nj][1]) : "v"(st + w_off[nj][1]) : "memory");
            asm volatile("ds_read_u8 %0, %1" : "=v"(scw[nj]) : "v"(st + w_sc[nj] + sb) : "memory");
        }
#pragma unroll
        for (int mj = 0; mj < 4; ++mj) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[mj][0]) : "v"(st + a_off[mj][0]) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[mj][1]) : "v"(st + a_off[mj][1]) : "memory");
            asm volatile("ds_read_u8 %0, %1" : "=v"(sca[mj]) : "v"(st + a_sc[mj] + sb) : "memory");
        }
        if (kt + 3 < nk) {
            issue_stage(kt + 3, (kt + 3) % X8_NST);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (X8_PIECES + 1)) : "memory");
        } else if (kt + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(X8_PIECES + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int mj = 0; mj < 4; ++mj) {
                const i32x8_t a = {(int)wf[nj][0].x, (int)wf[nj][0].y, (int)wf[nj][0].z, (int)wf[nj][0].w, (int)wf[nj][1].x, (int)wf[nj][1].y, (int)wf[nj][1].z, (int)wf[nj][1].w};
                const i32x8_t b = {(int)af[mj][0].x, (int)af[mj][0].y, (int)af[mj][0].z, (int)af[mj][0].w, (int)af[mj][1].x, (int)af[mj][1].y, (int)af[mj][1].z, (int)af[mj][1].w};
                acc[nj][mj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[nj][mj], 0, 0, 0, scw[nj], 0, sca[mj]);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp && kt + 1 == nk)) __builtin_amdgcn_s_barrier();
    }
    mx8_epilogue<OUT_F32, F>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// amax / quantise: HBM-bound streaming kernels, 16-byte accesses, grid-stride over rows
// ---------------------------------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void amax_kernel(const TX* __restrict__ x, int64_t ldx, int rows, int cols, float* __restrict__ amax) {
    constexpr int EPV = 16 / sizeof(TX);
    const int cpr = cols / EPV;  // 16-byte chunks per row
    const int64_t total = (int64_t)rows * cpr;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr);
        Vec16<TX> v;
        v.raw = ld16(x + r * ldx + c * EPV);
#pragma unroll
        for (int j = 0; j < EPV; ++j) m = fmaxf(m, fabsf(v.get(j)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));  // non-negative floats order as their bits
}

template <typename TX>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const TX* __restrict__ x, int64_t ldx, int rows, int cols, const float* __restrict__ amax,
                                                           unsigned char* __restrict__ y, int64_t ldy, float* __restrict__ scale_out) {
    // both quotients through fp64, i.e. correctly rounded fp32 values of 448 / amax and amax / 448 (the fp32 division
    // hipcc emits here was one ulp low, which moves every value that sits just past a rounding tie by one code)
    const float am = amax[0];
    const float inv = am > 0.f ? (float)(448.0 / (double)am) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) scale_out[0] = am > 0.f ? (float)((double)am / 448.0) : 1.0f;
    const int cpr = cols / 16;  // 16 outputs (16 bytes) per thread step
    const int64_t total = (int64_t)rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * 16;
        float f[16];
        if constexpr (sizeof(TX) == 2) {
            Vec16<TX> a, b;
            a.raw = ld16(x + r * ldx + c);
            b.raw = ld16(x + r * ldx + c + 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f[j] = a.get(j);
                f[8 + j] = b.get(j);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float4 t = *reinterpret_cast<const float4*>(x + r * ldx + c + 4 * h);
                f[4 * h] = t.x; f[4 * h + 1] = t.y; f[4 * h + 2] = t.z; f[4 * h + 3] = t.w;
            }
        }
        uint32_t w[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            // saturating round-to-nearest-even conversion of two floats into one half of a dword (v_cvt_pk_fp8_f32)
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f[4 * h] * inv, -448.f), 448.f), fminf(fmaxf(f[4 * h + 1] * inv, -448.f), 448.f), 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f[4 * h + 2] * inv, -448.f), 448.f), fminf(fmaxf(f[4 * h + 3] * inv, -448.f), 448.f), pk, true);
            w[h] = (uint32_t)pk;
        }
        st16(y + r * ldy + c, make_uint4(w[0], w[1], w[2], w[3]));
    }
}

// MXFP8 (OCP microscaling, e4m3 elements): one thread per 32-element block.  Block scale 2^e with the smallest e such that
// amax * 2^-e <= 448 (no element saturates), stored as E8M0 (e + 127) in byte (kb & 3) of dword [kb >> 2][row] of the scale
// array; elements are e4m3(x * 2^-e), round to nearest even.  A zero block gets the smallest scale (E8M0 0).
__device__ __forceinline__ void mx_block(const float (&f)[32], uint32_t (&w)[8], uint32_t& e8) {
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) am = fmaxf(am, fabsf(f[j]));
    const uint32_t bits = __float_as_uint(am);
    // amax = 1.m * 2^E: e = E - 8 if 1.m <= 1.75 (448 = 1.75 * 2^8) else E - 7; biased: (E + 127) - 8 [+ 1]
    int e = (int)(bits >> 23) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    e = e < 0 ? 0 : (e > 254 ? 254 : e);
    e8 = (uint32_t)e;
    const float inv = __uint_as_float((uint32_t)(254 - e) << 23);  // 2^(127 - e), exact
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f[4 * h] * inv, -448.f), 448.f), fminf(fmaxf(f[4 * h + 1] * inv, -448.f), 448.f), 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f[4 * h + 2] * inv, -448.f), 448.f), fminf(fmaxf(f[4 * h + 3] * inv, -448.f), 448.f), pk, true);
        w[h] = (uint32_t)pk;
    }
}

template <typename TX>
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const TX* __restrict__ x, int64_t ldx, int rows, int cols, unsigned char* __restrict__ y,
                                                             int64_t ldy, unsigned char* __restrict__ scales) {
    const int bpr = cols / 32;
    const int64_t total = (int64_t)rows * bpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / bpr;
        const int kb = (int)(i - r * bpr);
        const TX* src = x + r * ldx + kb * 32;
        float f[32];
        if constexpr (sizeof(TX) == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Vec16<TX> v;
                v.raw = ld16(src + 8 * q);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[8 * q + j] = v.get(j);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
                f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
            }
        }
        uint32_t w[8], e8;
        mx_block(f, w, e8);
        unsigned char* dst = y + r * ldy + kb * 32;
        st16(dst, make_uint4(w[0], w[1], w[2], w[3]));
        st16(dst + 16, make_uint4(w[4], w[5], w[6], w[7]));
        scales[((int64_t)(kb >> 2) * rows + r) * 4 + (kb & 3)] = (unsigned char)e8;
    }
}

}  // namespace lnxg
using namespace lnxg;

extern "C" int lnx_amax(const void* x, int x_dtype, int64_t ldx, int rows, int cols, float* amax, void* stream) {
    LNX_CHECK(x && amax && rows > 0 && cols > 0, "lnx_amax: null operand / empty");
    LNX_CHECK(x_dtype == LNX_F32 || x_dtype == LNX_BF16, "lnx_amax: bad dtype %d", x_dtype);
    const int epv = x_dtype == LNX_F32 ? 4 : 8;
    LNX_CHECK(cols % epv == 0 && ldx % epv == 0 && (((uintptr_t)x) & 15) == 0, "lnx_amax: 16-byte aligned rows of whole 16-byte chunks");
    int grid = (int)(((int64_t)rows * (cols / epv) + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (x_dtype == LNX_F32) hipLaunchKernelGGL(amax_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, rows, cols, amax);
    else hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, rows, cols, amax);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_quantize_fp8(const void* x, int x_dtype, int64_t ldx, int rows, int cols, const float* amax, void* y, int64_t ldy, float* scale_out, void* stream) {
    LNX_CHECK(x && amax && y && rows > 0 && cols > 0, "lnx_quantize_fp8: null operand / empty");
    LNX_CHECK(x_dtype == LNX_F32 || x_dtype == LNX_BF16, "lnx_quantize_fp8: bad dtype %d", x_dtype);
    LNX_CHECK(cols % 16 == 0 && ldx % 8 == 0 && ldy % 16 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0, "lnx_quantize_fp8: cols %% 16, 16-byte aligned rows");
    int grid = (int)(((int64_t)rows * (cols / 16) + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (x_dtype == LNX_F32)
        hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, rows, cols, amax, (unsigned char*)y, ldy, scale_out);
    else
        hipLaunchKernelGGL(quantize_fp8_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, rows, cols, amax, (unsigned char*)y, ldy, scale_out);
    LNX_LAUNCH_CHECK();
    return 0;
}

static int launch_fp8(const lnx_gemm_args* a, const float* a_scale, const float* w_scale, const void* mxa, const void* mxw, void* stream, const char* who) {
    LNX_CHECK(a != nullptr && a->A && a->W && a->C, "%s: null operand", who);
    LNX_CHECK(a->dtype == LNX_BF16, "%s: dtype (of C / c2 / aux) must be LNX_BF16", who);
    LNX_CHECK(a->M >= 256 && a->N > 0 && a->K >= 256 && a->K % 128 == 0, "%s: M=%d N=%d K=%d (M >= 256, K %% 128 == 0, K >= 256)", who, a->M, a->N, a->K);
    LNX_CHECK(a->lda % 16 == 0 && a->ldw % 16 == 0 && ((((uintptr_t)a->A) | ((uintptr_t)a->W)) & 15) == 0, "%s: A/W rows must be 16-byte aligned", who);
    LNX_CHECK(a->a_mode == LNX_ADDR_PLAIN && a->c_mode == LNX_ADDR_PLAIN && a->c_map.group == 0 && a->c_map.pad == 0 && a->c_map.off == 0,
              "%s: plain addressing only", who);
    // the fp8 epilogues carry GELU (+ pre-activation copy) and GELU' only: GELU_D / MUL_AUX / ReLU would be dispatched on the
    // feature mask of their bf16 twins and silently compute something else
    LNX_CHECK(a->act == LNX_ACT_NONE || a->act == LNX_ACT_GELU || a->act == LNX_ACT_GELU_BWD, "%s: act %d is not carried by the fp8 kernels (NONE, GELU, GELU_BWD)", who, a->act);
    if (a->act == LNX_ACT_GELU_BWD) LNX_CHECK(a->aux != nullptr, "%s: GELU_BWD needs aux", who);
    if (a->rowscale) LNX_CHECK(a->rows_per_sample > 0, "%s: rowscale needs rows_per_sample", who);
    const bool mx = mxa != nullptr;
    if (mx) LNX_CHECK(mxw != nullptr && ((((uintptr_t)mxa) | ((uintptr_t)mxw)) & 3) == 0, "%s: block-scale arrays must be 4-byte aligned", who);
    GemmP p;
    p.A = (const unsigned char*)a->A; p.W = (const unsigned char*)a->W; p.C = (unsigned char*)a->C; p.C2 = (unsigned char*)a->c2;
    p.aux = (const unsigned char*)a->aux; p.bias = a->bias; p.gamma = a->gamma; p.rowscale = a->rowscale; p.res = a->res;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldc2 = a->ldc2; p.ldaux = a->ldaux; p.ldres = a->ldres;
    p.M = a->M; p.N = a->N; p.K = a->K; p.a_mode = a->a_mode; p.c_mode = a->c_mode;
    p.pg = PatchGeom{a->Hin, a->Win, a->Cin};
    p.cmap = RowMap{0, 0, 0};
    p.act = a->act;
    p.rows_per_sample = a->rows_per_sample > 0 ? a->rows_per_sample : 1;
    p.tiles_m = cdiv(a->M, F8_BM);
    p.tiles_n = cdiv(a->N, F8_BN);
    p.sa = a_scale; p.sw = w_scale;
    p.mxa = (const uint32_t*)mxa; p.mxw = (const uint32_t*)mxw;
    const bool out_f32 = a->out_f32 != 0;
    int f = fast_epilogue_mask(p, out_f32);
    if (a->c8) {
        LNX_CHECK(mx && a->c8_scales && (f == (F_BIAS | F_C2 | F_GELU) || f == F_GELU_BWD) && a->N % 128 == 0 && a->ldc8 % 16 == 0 && (((uintptr_t)a->c8) & 15) == 0,
                  "%s: the MXFP8 output copy needs the MX kernel, the bias + GELU + pre-activation form or the GELU' form, N %% 128 == 0 and 16-byte aligned rows", who);
        p.C8 = (unsigned char*)a->c8; p.C8s = (unsigned char*)a->c8_scales; p.ldc8 = a->ldc8;
        f |= F_MXOUT;
    }
    LNX_CHECK(f != (int)F_GENERIC && a->gamma == nullptr, "%s: this epilogue needs the generic form, which the fp8 kernels do not carry", who);
    hipStream_t st = (hipStream_t)stream;
    {
        // the 256x256-tile MX kernel where it applies (LNX_FP8_X8=0: never)
        const char* e = getenv("LNX_FP8_X8");
        const bool x8_off = e && atoi(e) == 0;
        if (mx && !x8_off && a->N % X8_BN == 0 && a->K % 128 == 0 && a->K / X8_BK >= 4 && (int64_t)cdiv(a->M, X8_BM) * (a->N / X8_BN) >= 128) {
            p.tiles_m = cdiv(a->M, X8_BM);
            p.tiles_n = a->N / X8_BN;
            const int gridx = p.tiles_m * p.tiles_n;
            const size_t ldsx = X8_NST * (size_t)(X8_STAGE + X8_SCALES);
#define X8_LAUNCH(O, FF)                                                                                                              \
    do {                                                                                                                              \
        static bool attr = false;                                                                                                     \
        if (!attr) {                                                                                                                  \
            LNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_mx8_kernel<O, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx)); \
            attr = true;                                                                                                              \
        }                                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_mx8_kernel<O, FF>), dim3(gridx), dim3(512), ldsx, st, p);                                         \
    } while (0)
            bool done = true;
            if (out_f32 && f == (F_BIAS | F_RES)) X8_LAUNCH(true, F_BIAS | F_RES);
            else if (out_f32) done = false;
            else if (f == 0) X8_LAUNCH(false, 0);
            else if (f == F_BIAS) X8_LAUNCH(false, F_BIAS);
            else if (f == (F_BIAS | F_C2 | F_GELU) && a->act == LNX_ACT_GELU) X8_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
            else if (f == (F_BIAS | F_C2 | F_GELU | F_MXOUT) && a->act == LNX_ACT_GELU) X8_LAUNCH(false, F_BIAS | F_C2 | F_GELU | F_MXOUT);
            else if (f == F_GELU_BWD && a->act == LNX_ACT_GELU_BWD) X8_LAUNCH(false, F_GELU_BWD);
            else if (f == (F_GELU_BWD | F_MXOUT) && a->act == LNX_ACT_GELU_BWD) X8_LAUNCH(false, F_GELU_BWD | F_MXOUT);
            else done = false;
#undef X8_LAUNCH
            if (done) {
                note_nt_kernel(LNX_NT_KERNEL_MX8);
                LNX_LAUNCH_CHECK();
                return 0;
            }
            p.tiles_m = cdiv(a->M, F8_BM);
            p.tiles_n = cdiv(a->N, F8_BN);
        }
    }
    const int grid = p.tiles_m * p.tiles_n;
#define F8_LAUNCH_(O, FF, MXV)                                                                                                        \
    do {                                                                                                                              \
        const size_t lds = F8_NSTAGE * (size_t)(F8_STAGE + ((MXV) ? F8_SCALES : 0));                                                  \
        static bool attr = false;                                                                                                     \
        if (!attr) {                                                                                                                  \
            LNX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_fp8_kernel<O, FF, MXV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr = true;                                                                                                              \
        }                                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_fp8_kernel<O, FF, MXV>), dim3(grid), dim3(512), lds, st, p);                                      \
    } while (0)
#define F8_LAUNCH(O, FF)                 \
    do {                                 \
        if (mx) F8_LAUNCH_(O, FF, true); \
        else F8_LAUNCH_(O, FF, false);   \
    } while (0)
    if (out_f32) {
        LNX_CHECK(f == (F_BIAS | F_RES), "%s: an fp32 output needs bias + residual (the model's form)", who);
        F8_LAUNCH(true, F_BIAS | F_RES);
    } else if (f == 0) F8_LAUNCH(false, 0);
    else if (f == F_BIAS) F8_LAUNCH(false, F_BIAS);
    else if (f == (F_BIAS | F_C2 | F_GELU)) F8_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
    else if (f == (F_BIAS | F_C2 | F_GELU | F_MXOUT)) F8_LAUNCH_(false, F_BIAS | F_C2 | F_GELU | F_MXOUT, true);
    else if (f == F_GELU_BWD) F8_LAUNCH(false, F_GELU_BWD);
    else if (f == (F_GELU_BWD | F_MXOUT)) F8_LAUNCH_(false, F_GELU_BWD | F_MXOUT, true);
    else LNX_CHECK(false, "%s: unsupported epilogue feature set %d", who, f);
#undef F8_LAUNCH
#undef F8_LAUNCH_
    note_nt_kernel(LNX_NT_KERNEL_FP8);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_gemm_nt_fp8(const lnx_gemm_args* a, const float* a_scale, const float* w_scale, void* stream) {
    return launch_fp8(a, a_scale, w_scale, nullptr, nullptr, stream, "lnx_gemm_nt_fp8");
}

extern "C" int lnx_gemm_nt_mxfp8(const lnx_gemm_args* a, const void* a_scales, const void* w_scales, void* stream) {
    LNX_CHECK(a_scales != nullptr && w_scales != nullptr, "lnx_gemm_nt_mxfp8: null block-scale array");
    return launch_fp8(a, nullptr, nullptr, a_scales, w_scales, stream, "lnx_gemm_nt_mxfp8");
}

extern "C" int lnx_quantize_mxfp8(const void* x, int x_dtype, int64_t ldx, int rows, int cols, void* y, int64_t ldy, void* scales, void* stream) {
    LNX_CHECK(x && y && scales && rows > 0 && cols > 0, "lnx_quantize_mxfp8: null operand / empty");
    LNX_CHECK(x_dtype == LNX_F32 || x_dtype == LNX_BF16, "lnx_quantize_mxfp8: bad dtype %d", x_dtype);
    LNX_CHECK(cols % 128 == 0 && ldx % 8 == 0 && ldy % 16 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0 && (((uintptr_t)scales) & 3) == 0,
              "lnx_quantize_mxfp8: cols %% 128, 16-byte aligned rows, 4-byte aligned scales");
    const int64_t blocks = (int64_t)rows * (cols / 32);
    int grid = (int)((blocks + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (x_dtype == LNX_F32)
        hipLaunchKernelGGL(quantize_mxfp8_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, rows, cols, (unsigned char*)y, ldy, (unsigned char*)scales);
    else
        hipLaunchKernelGGL(quantize_mxfp8_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, rows, cols, (unsigned char*)y, ldy, (unsigned char*)scales);
    LNX_LAUNCH_CHECK();
    return 0;
}
