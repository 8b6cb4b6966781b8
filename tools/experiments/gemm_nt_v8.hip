// gemm_nt v8 (round 3): a persistent 256x128 kernel whose EPILOGUE RUNS ON ITS OWN WAVES.
//
// Why.  In the model the RoPE-block GEMMs move as many bytes in their epilogues as their K loops fetch (two [M, hidden] outputs
// of fc1, the fp32 residual read-modify-write of proj / fc2, the GELU' factor of the fc2 data gradient), and with every
// workgroup starting together the chip alternates between "all CUs in the K loop, HBM idle" and "all CUs in the epilogue,
// matrix cores idle" (the convoy).  Hiding the stores inside the next tile's K loop (gemm_nt_v7) only goes as
// far as one wave's in-order memory counter lets it: the wait for an LDS-DMA slice also waits for every store issued before
// it.  tools/ubench/store_overlap.hip shows the way out: while OTHER waves of the same workgroup saturate HBM with stores,
// a wave's LDS-DMA stream from L2 keeps its full rate (56.8 of 57.9 B/clk/CU).  So:
//
//   waves 0-3 (one per SIMD): the K loop and nothing else -- LDS-DMA ring of 32-element K slices (3 x 24 KiB), fragment
//       reads, 32 MFMAs per slice on a 128 x 64 wave tile (128 accumulator registers); their vmcnt only ever counts LDS-DMA.
//       The ring never drains (the last two iterations of a tile fetch the next tile's first slices).  At the end of a tile
//       the accumulators (bias folded into their initial value) are rounded to bf16 -- the Linear's output type under
//       autocast -- and handed over in a 64-KiB LDS image of the tile, row-major, 16-byte chunks XOR-swizzled so that both the
//       MFMA-layout writes and the row-layout reads are bank-conflict free.
//   waves 4-7: the epilogue of the PREVIOUS tile, a sixteenth at a time between the K loop's barriers: read 4 rows x 256 bytes
//       of the image, apply the epilogue (GELU and GELU' / GELU' multiply / DropPath-scaled fp32 residual add), store 16 bytes a
//       lane, whole 256-byte row segments per instruction.  Their stores (and operand fetches) sit in THEIR counter; the
//       K-loop waves never wait for them, and their VALU work runs beside the other wave's MFMAs on the same SIMD.
//
// One s_barrier per K iteration (all eight waves), two per tile for the hand-over.
#include "gemm_common.hpp"

namespace lnxg {

#define V8_READ4(dst, addr)                                                                             \
    do {                                                                                                \
        const uint32_t a_ = (addr);                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                        \
        asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst[1]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

constexpr int BM8 = 256, BN8 = 128, BK8 = 32, ROWB8 = 64;
constexpr int STAGE8 = (BM8 + BN8) * ROWB8;  // 24 KiB
constexpr int NST8 = 3;
constexpr int PIECES8 = STAGE8 / 1024 / 4;   // 1-KiB LDS-DMA instructions per K-loop wave and slice = 6
constexpr int IMG8 = BM8 * BN8 * 2;          // 64 KiB: the finished tile in bf16
constexpr int LDS8 = NST8 * STAGE8 + IMG8;
constexpr int UNITS8 = 16;                   // epilogue units (4 rows x 128 columns) per epilogue wave and tile

__device__ __forceinline__ int key8(int row) { return (row & 16) ? 3 : 0; }  // chunk key of an operand row (as gemm_nt_v4)
// chunk key of an image row: the 16 rows a 16-lane group of the K-loop waves writes at once ({0..3} + 16 {0..3} + const) get 16
// different keys, and so do ... a row's own 16 chunks for the readers
__device__ __forceinline__ int img_key(int row) { return (row & 3) | (((row >> 4) & 3) << 2); }

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));  // a register vector: usable as an inline-asm "+v" operand
__device__ __forceinline__ void v8_load16(u32x4_t& d, const void* ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr) : "memory"); }

// Diagnostic build only (-DV8_STAMP, tools/build_v8_stamp.sh): per-phase s_memtime sums of wave 0 (K loop) and wave 4 (epilogue)
// of one workgroup, written to the buffer LNX_V8_STAMPS names
#ifdef V8_STAMP
#define V8_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#define V8_T0() unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory")
#define V8_TOUT(base) do { unsigned long long* sp_ = reinterpret_cast<unsigned long long*>(p.C8s); \
                           if (sp_ && blockIdx.x == gridDim.x / 2 && lane == 0) for (int i_ = 0; i_ < 8; ++i_) sp_[(base) + i_] = tsum[i_]; } while (0)
#else
#define V8_T(i) do { } while (0)
#define V8_T0() do { } while (0)
#define V8_TOUT(base) do { } while (0)
#endif

template <bool OUT_F32, int F>
__global__ __launch_bounds__(512) void gemm_nt_v8_kernel(const GemmP p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3][A 256 rows | W 128 rows][64 B] | image [256][256 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK8;  // >= 4
    unsigned char* img = smem + NST8 * STAGE8;
    auto tile_origin = [&](int tile, int& m0, int& n0) __attribute__((always_inline)) {
        const int logical = xcd_remap(tile, ntiles);
        n0 = (logical % p.tiles_n) * BN8;
        m0 = (logical / p.tiles_n) * BM8;
    };
    int tile = blockIdx.x, m0, n0;
    if (tile >= ntiles) return;
    tile_origin(tile, m0, n0);

    if (wave < 4) {
        // ======================================= K-loop waves =======================================
        const int wm = wave & 1, wn = wave >> 1;  // 128-row half, 64-column half of the tile
        const int s = lane & 15, g = lane >> 4;
        const int frag_row = (s >> 2) * 16 + (s & 3);
        const uint32_t chunk_off = (uint32_t)((g ^ (((s >> 2) & 1) * 3)) << 4);
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const uint32_t a_off = (uint32_t)((wm * 128 + frag_row) * ROWB8) + chunk_off;
        const uint32_t w_off = (uint32_t)((BM8 + wn * 64 + frag_row) * ROWB8) + chunk_off;
        typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
        u32x8_t src = {0, 0, 0, 0, 0, 0, 0, 0}, srcn = {0, 0, 0, 0, 0, 0, 0, 0};
        auto piece_offset = [&](int j, int tm0, int tn0) __attribute__((always_inline)) -> uint32_t {
            const int i = wave + 4 * j;
            const int row = 16 * i + (lane >> 2);
            const int slot = lane & 3;
            if (row < BM8) {
                int m = tm0 + row;
                if (m >= p.M) m = p.M - 1;
                return (uint32_t)(m * (int)p.lda + (slot ^ key8(row)) * 8) * 2u;
            }
            const int wr = row - BM8;
            int n = tn0 + wr;
            if (n >= p.N) n = p.N - 1;
            return (uint32_t)(n * (int)p.ldw + (slot ^ key8(wr)) * 8) * 2u;
        };
        auto issue = [&](bool of_next, int kslice, int stage) __attribute__((always_inline)) {
            const u32x8_t sv = of_next ? srcn : src;
#pragma unroll
            for (int j = 0; j < PIECES8; ++j) {
                const int i = wave + 4 * j;
                const unsigned char* base = 16 * i < BM8 ? p.A : p.W;  // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (sv[j] + (uint32_t)(kslice * BK8 * 2))),
                                                 (__attribute__((address_space(3))) void*)(smem + stage * STAGE8 + i * 1024), 16, 0, 0);
            }
        };
        // bias of this lane's 16 columns (n0 + wn 64 + 16 g ..): the accumulators start from it.  Fetched by untracked loads:
        // for the first tile before the ring's prologue, for every later one three iterations before the end of its
        // predecessor, in front of that iteration's LDS-DMA pieces -- the counted waits of the K loop cover them.
        u32x4_t bias4[4];
        (void)bias4;
        auto fetch_bias = [&](int tn0) __attribute__((always_inline)) {
            int nb = tn0 + wn * 64 + g * 16;
            if (nb >= p.N) nb = 0;  // columns beyond N: anything valid (never stored)
#pragma unroll
            for (int h = 0; h < 4; ++h) v8_load16(bias4[h], p.bias + nb + 4 * h);
        };
#pragma unroll
        for (int j = 0; j < PIECES8; ++j) src[j] = piece_offset(j, m0, n0);
        auto issue_piece = [&](bool of_next, int kslice, int stage, int j) __attribute__((always_inline)) {  // j: compile-time at every call site
            const int i = wave + 4 * j;
            const unsigned char* base = 16 * i < BM8 ? p.A : p.W;
            const uint32_t so = of_next ? srcn[j] : src[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (so + (uint32_t)(kslice * BK8 * 2))),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE8 + i * 1024), 16, 0, 0);
        };
        // The software pipeline of one K-loop wave.  At the top of iteration kt: the fragments of slice kt are in registers, slice
        // kt + 1 is complete in LDS (stage r + 1), slice kt + 2 is on its way (stage r + 2), stage r is free.  The iteration
        // multiplies slice kt while it reads slice kt + 1's fragments -- each A fragment into the registers of the one the last
        // four MFMAs have just consumed, the W fragments into a second set -- and issues slice kt + 3 into stage r, one LDS-DMA
        // piece between MFMA groups: L2 feed, LDS reads and the matrix pipe run together instead of one after the other
        // (measured on the first version: 474 + 339 + 577 cycles per iteration in sequence).
        if (F & F_BIAS) fetch_bias(n0);
        issue(false, 0, 0);
        issue(false, 1, 1);
        issue(false, 2, 2);
        uint4 wfa[4], wfb[4], af[8];
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES8) : "memory");  // slice 0 (and the bias) landed
        __builtin_amdgcn_s_barrier();
        V8_READ4(wfa, lds_base + w_off);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(lds_base + a_off), "i"((i >> 2) * 64 * ROWB8 + (i & 3) * 256) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES8) : "memory");  // slice 1 landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int ring = 0;  // stage of the slice whose fragments are in registers
        V8_T0();

        while (true) {
            const int next = tile + gridDim.x;
            const bool has_next = next < ntiles;
            int nm0 = 0, nn0 = 0;
            if (has_next) {
                tile_origin(next, nm0, nn0);
#pragma unroll
                for (int j = 0; j < PIECES8; ++j) srcn[j] = piece_offset(j, nm0, nn0);
            }
            f32x4_t acc0[4][4], acc1[4][4];  // [ni][mi]: rows 0..63 / 64..127 of the wave tile
            if (F & F_BIAS) {
                // the bias registers were written by untracked loads; every counted wait since lies behind us HERE, and this
                // (empty, volatile) statement is what keeps the compiler from reading them any earlier
#pragma unroll
                for (int h = 0; h < 4; ++h) asm volatile("" : "+v"(bias4[h]));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4_t b = {0.f, 0.f, 0.f, 0.f};
                if (F & F_BIAS) b = f32x4_t{__uint_as_float(bias4[ni][0]), __uint_as_float(bias4[ni][1]), __uint_as_float(bias4[ni][2]), __uint_as_float(bias4[ni][3])};
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc0[ni][mi] = b;
                    acc1[ni][mi] = b;
                }
            }
            auto kiter = [&](const int kt, uint4 (&wfc)[4], uint4 (&wfn)[4]) __attribute__((always_inline)) {
                const int r1 = ring == 2 ? 0 : ring + 1;
                const uint32_t stn = lds_base + r1 * STAGE8;
                // what goes into the freed stage: slice kt + 3 of this tile, of the next one, or nothing (the last tile's tail)
                const int ks = kt + 3;
                const bool cur = ks < nk, any = cur || has_next;
                if ((F & F_BIAS) && kt == nk - 4 && has_next) fetch_bias(nn0);  // in front of this iteration's pieces: its closing wait covers them
                V8_READ4(wfn, stn + w_off);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if (i < 4) Mfma<T>::run(acc0[ni][i], wfc[ni], af[i]);
                        else Mfma<T>::run(acc1[ni][i - 4], wfc[ni], af[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(stn + a_off), "i"((i >> 2) * 64 * ROWB8 + (i & 3) * 256) : "memory");
                    if (i < PIECES8 && any) issue_piece(!cur, cur ? ks : ks - nk, ring, i);
                }
                __builtin_amdgcn_sched_barrier(0);
                V8_T(0);  // MFMAs, fragment reads and LDS-DMA pieces issued
                if (any) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES8) : "memory");  // own pieces of slice kt + 2 landed
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                V8_T(1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                V8_T(2);
                __builtin_amdgcn_s_barrier();
                V8_T(3);
                ring = r1;
            };
            for (int kt = 0; kt < nk; kt += 2) {  // nk is even: the W fragment sets swap roles every iteration
                kiter(kt, wfa, wfb);
                kiter(kt + 1, wfb, wfa);
            }
            // ---- hand-over: lane (s, g) holds, for row slot mi, columns wn 64 + 16 g .. + 15 of row wm 128 + a 64 + frag_row + 4 mi
            __builtin_amdgcn_s_barrier();  // the epilogue waves are done with the previous image
            V8_T(5);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int row = wm * 128 + a * 64 + frag_row + 4 * mi;  // img_key(row) == s
                    Vec16<T> o[2];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[ni >> 1].set((ni & 1) * 4 + r, a == 0 ? acc0[ni][mi][r] : acc1[ni][mi][r]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) st16(img + row * 256 + (((wn * 8 + 2 * g + h) ^ s) << 4), o[h].raw);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // the image is complete
            V8_T(6);
            if (!has_next) break;
            tile = next;
            m0 = nm0;
            n0 = nn0;
            src = srcn;
        }
        if (wave == 0) V8_TOUT(0);
        return;
    }

    // ======================================= epilogue waves =======================================
    const int ew = wave - 4;                    // rows 64 ew .. 64 ew + 63 of the image
    const int rl = lane >> 4, cl = lane & 15;   // row within a unit, 16-byte chunk (8 columns) of the row
    // one unit = 4 rows x 128 columns of the finished tile at (pm0, pn0)
    auto unit = [&](int u, int pm0, int pn0) __attribute__((always_inline)) {
        const int row = 64 * ew + 4 * u + rl;
        const int m = pm0 + row, n = pn0 + 8 * cl;
        const bool ok = m < p.M && n < p.N;
        const int mc = min(m, p.M - 1), nc = n < p.N ? n : 0;
        // operand fetches first (unconditional, clamped)
        uint4 ax = make_uint4(0u, 0u, 0u, 0u);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        float rsc = 1.0f;
        if (F & F_GELU_BWD) ax = ld16(reinterpret_cast<const T*>(p.aux) + ((int64_t)mc * p.ldaux + nc));
        if (F & F_RES) {
            const float* rp = p.res + ((int64_t)mc * p.ldres + nc);
            r0 = *reinterpret_cast<const float4*>(rp);
            r1 = *reinterpret_cast<const float4*>(rp + 4);
            if (p.rowscale) rsc = p.rowscale[mc / p.rows_per_sample];
        }
        Vec16<T> t;
        t.raw = ld16(img + row * 256 + ((cl ^ img_key(row)) << 4));
        if (F == 0 || F == F_BIAS) {
            if (ok) st16(reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + n), t.raw);
            return;
        }
        f32x2_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = f32x2_t{t.get(2 * i), t.get(2 * i + 1)};
        if ((F & F_GELU) && (F & F_C2)) {  // fc1: C = GELU(v); C2 = v, or GELU'(v) for the GELU_D form (kernel-uniform)
            f32x2_t a[4], d[4];
            Vec16<T> oc, o2;
            if (p.act == LNX_ACT_GELU_D) {
                gelu_lean_grad2_n<4>(v, a, d);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o2.set(2 * i, d[i].x);
                    o2.set(2 * i + 1, d[i].y);
                }
            } else {
                gelu_lean2_n<4>(v, a);
                o2.raw = t.raw;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                oc.set(2 * i, a[i].x);
                oc.set(2 * i + 1, a[i].y);
            }
            if (ok) {
                st16(reinterpret_cast<T*>(p.C2) + ((int64_t)m * p.ldc2 + n), o2.raw);
                st16(reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + n), oc.raw);
            }
            return;
        }
        if (F & F_GELU_BWD) {
            Vec16<T> xa;
            xa.raw = ax;
            if (p.act == LNX_ACT_MUL_AUX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] * f32x2_t{xa.get(2 * i), xa.get(2 * i + 1)};
            } else {
                f32x2_t x[4], a[4], d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = f32x2_t{xa.get(2 * i), xa.get(2 * i + 1)};
                gelu_lean_grad2_n<4>(x, a, d);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] * d[i];
            }
            Vec16<T> o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o.set(2 * i, v[i].x);
                o.set(2 * i + 1, v[i].y);
            }
            if (ok) st16(reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + n), o.raw);
            return;
        }
        if (F & F_RES) {
            if (ok) {
                float* cp = reinterpret_cast<float*>(p.C) + ((int64_t)m * p.ldc + n);
                *reinterpret_cast<float4*>(cp) = make_float4(fmaf(v[0].x, rsc, r0.x), fmaf(v[0].y, rsc, r0.y), fmaf(v[1].x, rsc, r0.z), fmaf(v[1].y, rsc, r0.w));
                *reinterpret_cast<float4*>(cp + 4) = make_float4(fmaf(v[2].x, rsc, r1.x), fmaf(v[2].y, rsc, r1.y), fmaf(v[3].x, rsc, r1.z), fmaf(v[3].y, rsc, r1.w));
            }
        }
    };

    bool have_prev = false;
    int pm0 = 0, pn0 = 0;
    __builtin_amdgcn_s_barrier();  // partners of the K-loop waves' two prologue barriers
    __builtin_amdgcn_s_barrier();
    V8_T0();
    while (true) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        for (int kt = 0; kt < nk; ++kt) {
            if (have_prev) {
                const int u_lo = kt * UNITS8 / nk, u_hi = (kt + 1) * UNITS8 / nk;  // the previous tile's epilogue, spread over this K loop
                for (int u = u_lo; u < u_hi; ++u) unit(u, pm0, pn0);
            }
            V8_T(0);  // epilogue units
            __builtin_amdgcn_s_barrier();
            V8_T(1);  // iteration barrier
        }
        __builtin_amdgcn_s_barrier();  // done with the previous image
        __builtin_amdgcn_s_barrier();  // the new image is complete
        V8_T(2);
        have_prev = true;
        pm0 = m0;
        pn0 = n0;
        if (!has_next) break;
        tile = next;
        tile_origin(tile, m0, n0);
    }
    for (int u = 0; u < UNITS8; ++u) unit(u, pm0, pn0);  // the last tile of this workgroup
    V8_T(3);
    if (wave == 4) V8_TOUT(8);
}

bool nt_v8_ok(const GemmP& p, int f, bool out_f32) {
    if (f == (int)F_GENERIC || p.a_mode == LNX_ADDR_PATCH2) return false;
    if (p.K % (2 * BK8) != 0 || p.K / BK8 < 4) return false;  // an even number of K slices (the W fragment sets alternate)
    const int64_t lim = (int64_t)1 << 31;  // 32-bit byte offsets of the LDS-DMA sources
    if ((int64_t)p.M * p.lda * 2 >= lim || (int64_t)p.N * p.ldw * 2 >= lim) return false;
    if (out_f32) return f == (F_BIAS | F_RES);
    return f == 0 || f == F_BIAS || f == (F_BIAS | F_C2 | F_GELU) || f == F_GELU_BWD;
}

int launch_nt_v8(const GemmP& p0, int f, bool out_f32, hipStream_t st) {
    GemmP p = p0;
    p.tiles_m = cdiv(p.M, BM8);
    p.tiles_n = cdiv(p.N, BN8);
    const int ntiles = p.tiles_m * p.tiles_n;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        cus = prop.multiProcessorCount;
    }
    const int grid = ntiles < cus ? ntiles : cus;  // persistent: one workgroup per CU
#ifdef V8_STAMP
    const char* sp = getenv("LNX_V8_STAMPS");  // device address of a 16 x uint64 buffer
    p.C8s = sp ? reinterpret_cast<unsigned char*>(strtoull(sp, nullptr, 10)) : nullptr;
#endif
#define V8_LAUNCH(O, FF)                                                                                                              \
    do {                                                                                                                              \
        static bool attr = false;                                                                                                     \
        if (!attr) {                                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v8_kernel<O, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8); \
            attr = true;                                                                                                              \
        }                                                                                                                             \
        hipLaunchKernelGGL((gemm_nt_v8_kernel<O, FF>), dim3(grid), dim3(512), LDS8, st, p);                                           \
    } while (0)
    if (out_f32) V8_LAUNCH(true, F_BIAS | F_RES);
    else if (f == 0) V8_LAUNCH(false, 0);
    else if (f == F_BIAS) V8_LAUNCH(false, F_BIAS);
    else if (f == (F_BIAS | F_C2 | F_GELU)) V8_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
    else V8_LAUNCH(false, F_GELU_BWD);
#undef V8_LAUNCH
    return 0;
}

}  // namespace lnxg
