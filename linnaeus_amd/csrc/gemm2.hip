// gemm_nt v2: 256x128 output tile, 8 waves (4x2, each the same 64x64 / 4x4-MFMA wave tile as
// gemm.hip), bf16 only, K % 64 == 0.
//
// What changes against the 128x128 register-staged kernel is the memory pipeline.  That kernel is
// bound by global-load latency: one tile of prefetch in VGPRs cannot cover ~2 us of loaded-chip
// latency with 512 cycles of MFMA work.  Here operand tiles go HBM/L2 -> LDS directly
// (global_load_lds_dwordx4: no VGPRs, no ds_write pass) into a 3-deep ring of 48 KiB stages, two
// K tiles always in flight behind a COUNTED s_waitcnt vmcnt(6) and a raw s_barrier (a
// __syncthreads() would drain the queue with vmcnt(0)).  One barrier per K tile:
//
//   iteration t:  wait(tile t landed) ; barrier ; issue tile t+2 -> stage (t+2)%3 ; MFMA on stage t%3
//
// Stage (t+2)%3 was last read in iteration t-1, and every wave has consumed those reads (its MFMAs
// need them) before it can arrive at this iteration's barrier, so the refill is WAR-safe.
// LDS-DMA writes are lane-linear (wave-uniform base + lane*16 B), so the bank-conflict swizzle is
// applied on the per-lane GLOBAL source address: LDS slot (row, c) receives global chunk
// c ^ key(row), and fragment reads use the same key (gemm.hip's conflict-free permutation).
// Rows beyond M / N are clamped to the last valid row (their results are never stored).
#include "gemm_common.hpp"

namespace lnxg {

// four 16-byte fragment reads at rows +0, +4, +8, +12 (byte offsets 0/512/1024/1536) from one address
#define DS_READ4(dst, addr)                                                                              \
    do {                                                                                                 \
        const uint32_t a_ = (addr);                                                                      \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                         \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[1]) : "v"(a_) : "memory");              \
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:1536" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

constexpr int BM2 = 256, BN2 = 128;
constexpr int STAGE_BYTES = (BM2 + BN2) * ROWB;  // 48 KiB
constexpr int NSTAGE = 3;
constexpr int LD_PER_WAVE = (BM2 + BN2) / 8 / 8;  // 1 KiB wave-instructions per wave per stage = 6

// F = feature set of a specialised epilogue (gemm_common.hpp) or F_GENERIC
template <bool OUT_F32, bool PATCH, int F>
__global__ __launch_bounds__(512) void gemm_nt_v2_kernel(const GemmP p) {
    typedef bf16_t T;
    constexpr int BK = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NSTAGE][A 256 rows | W 128 rows][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int s = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // ---- LDS-DMA source addresses: wave w issues wave-instructions i = w + 8 j (j < 6); instruction
    // i fills LDS rows 8i..8i+7 of the stage (rows 0..255 = A, 256..383 = W), lane -> (row l>>3, slot l&7)
    const unsigned char* src[LD_PER_WAVE];
#pragma unroll
    for (int j = 0; j < LD_PER_WAVE; ++j) {
        const int i = wave + 8 * j;
        const int row = 8 * i + (lane >> 3);
        const int slot = lane & 7;
        if (row < BM2) {
            const int chunk = slot ^ row_key(row);
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            const int64_t base = PATCH ? patch_base(p.pg, m) : (int64_t)m * p.lda;
            // PATCH2: a 16-byte chunk never straddles a (kh) segment because 2*Cin % 8 == 0
            src[j] = p.A + (base + chunk * 8) * 2;
        } else {
            const int wr = row - BM2;
            const int chunk = slot ^ row_key(wr);
            int n = n0 + wr;
            if (n >= p.N) n = p.N - 1;
            src[j] = p.W + ((int64_t)n * p.ldw + chunk * 8) * 2;
        }
    }
    auto issue_tile = [&](int kt, int stage) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < LD_PER_WAVE; ++j) {
            const int i = wave + 8 * j;
            const unsigned char* gp = src[j];
            if (PATCH && 8 * i < BM2) {
                // element offset k -> patch_col(k): add the row jump of the (kh) segment this chunk is in
                const int slot = lane & 7;
                const int row = 8 * i + (lane >> 3);
                const int kk = k0 + ((slot ^ row_key(row)) * 8);
                gp = src[j] + (patch_col(p.pg, kk) - ((slot ^ row_key(row)) * 8)) * 2;
            } else {
                gp += (int64_t)k0 * 2;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const int frag_key = ((s >> 1) & 1) | ((s >> 2) << 1);
    const int a_row0 = wm * 64 + frag_row;
    const int w_row0 = BM2 + wn * 64 + frag_row;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t ch0 = (uint32_t)((g ^ frag_key) << 4), ch1 = (uint32_t)(((g + 4) ^ frag_key) << 4);
    const uint32_t a_off0 = a_row0 * ROWB + ch0, a_off1 = a_row0 * ROWB + ch1;
    const uint32_t w_off0 = w_row0 * ROWB + ch0, w_off1 = w_row0 * ROWB + ch1;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;  // >= 2
    issue_tile(0, 0);
    issue_tile(1, 1);
    // Ping-pong schedule: waves 0-3 and 4-7 (one of each per SIMD) run half a K slice apart, so one wave's
    // memory phase (fragment reads + LDS-DMA issue, ~60+ cycles of issue per 1-KiB piece) sits under its SIMD
    // partner's 32 MFMAs instead of every wave doing both phases in lock step.  Two barriers per K slice:
    //   group 0:      mem(0) | comp(0) | mem(1) | comp(1) | ...
    //   group 1:  --  |  mem(0) | comp(0) | mem(1) | ...
    // Each wave ends its memory phase k by waiting (counted vmcnt) for its own pieces of stage k+1 and for its
    // fragment reads; the barrier that follows therefore publishes stage k+1 and retires the reads of stage k,
    // whose slot is refilled in memory phase k+1 (by stage k+3).  Fragment reads are inline asm on purpose: with
    // compiler-visible LDS loads hipcc puts an s_waitcnt vmcnt(0) in front of them (it must assume they alias the
    // in-flight LDS-DMA writes), which would drain the prefetch every iteration.
    const int grp = wave >> 2;
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t st = lds_base + (kt % NSTAGE) * STAGE_BYTES;
        uint4 wf0[4], af0[4], wf1[4], af1[4];
        DS_READ4(wf0, st + w_off0);
        DS_READ4(af0, st + a_off0);
        DS_READ4(wf1, st + w_off1);
        DS_READ4(af1, st + a_off1);
        if (kt + 2 < nk) {
            issue_tile(kt + 2, (kt + 2) % NSTAGE);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
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
            for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc[ni][mi], wf0[ni], af0[mi]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc[ni][mi], wf1[ni], af1[mi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp && kt + 1 == nk)) __builtin_amdgcn_s_barrier();
    }
    if (F == F_GENERIC) gemm_epilogue<T, OUT_F32>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
    else gemm_epilogue_fast<T, OUT_F32, F>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------
// gemm_nt v4: 256x256 output tile for problems whose N is a multiple of 256.  The K loop of the 256x128 kernel is
// bound by the LDS-DMA fill rate (measured ~32 B/clk/CU; it needs 47 B/clk to keep the MFMA pipe full): a 256x256
// tile moves 1.5x fewer operand bytes per FLOP, 32 B/clk at full MFMA rate.  Eight waves, each a 128x64 wave tile
// (two 4x4 accumulator sets, 128 registers); K in 32-element slices (64-byte LDS rows: 512 rows = 32 KiB per stage)
// through a 4-deep ring with three slices in flight; same ping-pong wave groups and specialised epilogues.
// 64-byte rows keep global chunk c at slot c ^ key, key = 3 * bit4(row): with it the permuted fragment rows
// {0..3,16..19,32..35,48..51}+4i of every 16-lane group of a ds_read_b128 fall on 16 distinct 16-byte bank slots.
// ------------------------------------------------------------------------------------
#define DS4_READ4(dst, addr)                                                                            \
    do {                                                                                                \
        const uint32_t a_ = (addr);                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                        \
        asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst[1]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

constexpr int BM4 = 256, BN4 = 256, BK4 = 32, ROWB4 = 64;
constexpr int STAGE4 = (BM4 + BN4) * ROWB4;   // 32 KiB
constexpr int NST4 = 4;
constexpr int PIECES4 = STAGE4 / 1024 / 8;    // 1-KiB LDS-DMA instructions per wave per stage = 4

__device__ __forceinline__ int key4r(int row) { return (row & 16) ? 3 : 0; }

template <bool OUT_F32, int F>
__global__ __launch_bounds__(512) void gemm_nt_v4_kernel(const GemmP p) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST4][A 256 rows | W 256 rows][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // waves 0-3 / 4-7 (the ping-pong groups) cover the column halves
    const int s = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * BM4, n0 = tn * BN4;

    // piece i (32 per stage) fills LDS rows 16 i .. 16 i + 15 (rows 0..255 = A, 256..511 = W); wave w issues i = w + 8 j
    const unsigned char* src[PIECES4];
#pragma unroll
    for (int j = 0; j < PIECES4; ++j) {
        const int i = wave + 8 * j;
        const int row = 16 * i + (lane >> 2);
        const int slot = lane & 3;
        if (row < BM4) {
            const int chunk = slot ^ key4r(row);
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[j] = p.A + ((int64_t)m * p.lda + chunk * 8) * 2;
        } else {
            const int wr = row - BM4;
            const int chunk = slot ^ key4r(wr);
            int n = n0 + wr;
            if (n >= p.N) n = p.N - 1;
            src[j] = p.W + ((int64_t)n * p.ldw + chunk * 8) * 2;
        }
    }
    auto issue_stage = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < PIECES4; ++j) {
            const int i = wave + 8 * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (int64_t)kt * BK4 * 2),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE4 + i * 1024), 16, 0, 0);
        }
    };

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const uint32_t chunk_off = (uint32_t)((g ^ (((s >> 2) & 1) * 3)) << 4);  // key of every row this lane reads = 3 * bit0(s >> 2)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = (uint32_t)((wm * 128 + frag_row) * ROWB4) + chunk_off;
    const uint32_t w_off = (uint32_t)((BM4 + wn * 64 + frag_row) * ROWB4) + chunk_off;

    f32x4_t acc0[4][4], acc1[4][4];  // rows 0..63 / 64..127 of the wave tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

    const int nk = p.K / BK4;  // >= 4
    issue_stage(0, 0);
    issue_stage(1, 1);
    issue_stage(2, 2);
    const int grp = wave >> 2;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // own pieces of slice 0 have landed (two slices = 8 pieces still in flight)
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t st = lds_base + (kt % NST4) * STAGE4;
        uint4 wf[4], af0[4], af1[4];
        DS4_READ4(wf, st + w_off);
        DS4_READ4(af0, st + a_off);
        DS4_READ4(af1, st + a_off + 64 * ROWB4);
        // end of memory phase kt: own pieces of slice kt+1 landed, fragment reads of slice kt done; the slot refilled
        // in memory phase kt+1 (slice kt+4) is the one read in phase kt
        if (kt + 3 < nk) {
            issue_stage(kt + 3, (kt + 3) % NST4);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (kt + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
            for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc0[ni][mi], wf[ni], af0[mi]);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc1[ni][mi], wf[ni], af1[mi]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp && kt + 1 == nk)) __builtin_amdgcn_s_barrier();
    }
    if (F == F_GENERIC) {
        gemm_epilogue<T, OUT_F32>(p, acc0, m0 + wm * 128, n0 + wn * 64, lane);
        gemm_epilogue<T, OUT_F32>(p, acc1, m0 + wm * 128 + 64, n0 + wn * 64, lane);
    } else {
        gemm_epilogue_fast<T, OUT_F32, F>(p, acc0, m0 + wm * 128, n0 + wn * 64, lane);
        gemm_epilogue_fast<T, OUT_F32, F>(p, acc1, m0 + wm * 128 + 64, n0 + wn * 64, lane);
    }
}

nt_experiment_fn g_nt_experiment = nullptr;

// default choice (LNX_NT_V7 unset): from the measurements of tools/bench_gemm_epi.py
// Measured (profiles/r03_nt_v7.log, sm shapes at B = 256): against the one-shot 256x128 kernel the persistent kernel wins
// 10-30 % when it has at least two tiles per CU (no pipeline refill per tile, stores spread over the K loop) and the epilogue is
// light (plain / bias / fp32 residual); it ties with the 256x256 kernel where that one applies (N % 256 == 0) and with the GELU
// forms (their epilogue arithmetic, not their stores, is what the K loop waits for), and loses a few percent below two tiles
// per CU.
// Round 4: 256x256 or 256x128 tiles for a product whose N allows both?  Every tile of a launch costs the same, so a launch takes
// ceil(tiles / CUs) rounds whatever the scheduler: 600 tiles of 256x256 on 256 CUs are three rounds for 2.3 rounds of work (the
// N = 1536 products of BASELINE config 3's 128 images per GPU), the same product in 1200 tiles of 256x128 five half-size rounds.  The
// big tile moves 1.5x fewer operand bytes per FLOP, which is worth ~3 % at K = 384 and ~12 % on long K loops (profiles/r03_bare_gemm.log).
// LNX_NT_TILE_COST=0: the round-3 rule (big tile wherever N % 256 == 0).
static bool big_tile_wins(const GemmP& p) {
    static const bool off = getenv("LNX_NT_TILE_COST") && atoi(getenv("LNX_NT_TILE_COST")) == 0;
    if (off) return true;
    const int dc = device_cus();
    const int cus = persistent_cus(dc > 0 ? dc : 256);
    const int64_t rows = cdiv(p.M, 256);
    const double big = (double)cdiv(rows * (p.N / BN4), (int64_t)cus) * 2.0 * (p.K <= 512 ? 0.97 : 0.88);
    const double small = (double)cdiv(rows * cdiv(p.N, 128), (int64_t)cus);
    return small >= 0.93 * big;  // (the small tile has to win by a margin: measured, a tie on paper goes to the big tile at sm / lg / xl)
}

static bool nt_v7_preferred(const GemmP& p, int f, bool out_f32) {
    const bool two_per_cu = (int64_t)cdiv(p.M, 256) * cdiv(p.N, 128) >= 512;
    const bool big_better = p.N % BN4 == 0 && big_tile_wins(p);
    if (f == F_GELU_BWD && p.act == LNX_ACT_MUL_AUX) return two_per_cu;  // 121.6 -> 115.7 us at the sm fc2 data gradient, also against the 256x256 tile
    if (f == F_GELU_BWD || f == (F_BIAS | F_C2 | F_GELU)) return two_per_cu && p.N % BN4 == 0 && !big_better;  // only instead of a badly filling big tile
    if (f != 0 && f != F_BIAS && !(out_f32 && f == (F_BIAS | F_RES))) return false;
    if (big_better) return false;  // the 256x256 tile (half the fill traffic per FLOP) is the better kernel there
    return two_per_cu;
}

static bool nt_v4_ok(const GemmP& p, int f) {
    static const bool off = getenv("LNX_NT_V4") && atoi(getenv("LNX_NT_V4")) == 0;  // A/B switch for benchmarking
    if (off || f == (int)F_GENERIC || p.a_mode == LNX_ADDR_PATCH2) return false;
    return p.N % BN4 == 0 && p.K % BK4 == 0 && p.K >= 4 * BK4;
}

bool nt_v2_ok(const GemmP& p, int dtype) {
    if (dtype != LNX_BF16) return false;
    if (p.K % 64 != 0 || p.K < 128) return false;
    if (p.M < 1024) return false;  // tiny-M GEMMs (tail, meta heads) stay on the 128x128 kernel
    return true;
}

// Which pipelined family a product takes (no launch: lnx_nt_dispatch and launch_nt_v2 share this) -- the rules of DESIGN.md's dispatch table:
//   LNX_NT_V7: 1 = every shape the persistent deferred-store kernel can run, 0 = never, unset = the measured choice (nt_v7_preferred)
//   LNX_NT_V9: 1 = the persistent 256x256 kernel wherever it can run, 0 = never, unset = with at least 1.5 tiles per CU (below that a
//              workgroup has no second tile to hide the first one's epilogue under) and where the big tile wins on rounds
int nt_v2_family(const GemmP& p, bool out_f32, int* f_out) {
    const bool patch = p.a_mode == LNX_ADDR_PATCH2;
    static const bool no_fast = getenv("LNX_NT_GENERIC_EPI") != nullptr;  // A/B switch for benchmarking
    const int f = (patch || no_fast) ? (int)F_GENERIC : fast_epilogue_mask(p, out_f32);
    if (f_out) *f_out = f;
    {
        const char* e7 = getenv("LNX_NT_V7");
        const int v7 = e7 ? atoi(e7) : -1;
        if (v7 != 0 && nt_v7_ok(p, f, out_f32) && (v7 == 1 || nt_v7_preferred(p, f, out_f32))) return LNX_NT_KERNEL_V7;
    }
    const char* e9 = getenv("LNX_NT_V9");
    const int v9 = e9 ? atoi(e9) : -1;
    if (nt_v4_ok(p, f) && (v9 == 1 || big_tile_wins(p))) {
        const int dc = device_cus();
        const int64_t cus = dc > 0 ? dc : 256;
        if (v9 != 0 && nt_v9_ok(p, f, out_f32) && (v9 == 1 || (int64_t)cdiv(p.M, BM4) * (p.N / BN4) * 2 >= 3 * cus)) return LNX_NT_KERNEL_V9;
        return LNX_NT_KERNEL_V4;
    }
    return LNX_NT_KERNEL_V2;
}

int launch_nt_v2(const GemmP& p0, bool out_f32, hipStream_t st) {
    GemmP p = p0;
    p.tiles_m = cdiv(p.M, BM2);
    p.tiles_n = cdiv(p.N, BN2);
    const int grid = p.tiles_m * p.tiles_n;
    const size_t lds = NSTAGE * STAGE_BYTES;
    const bool patch = p.a_mode == LNX_ADDR_PATCH2;
    int f = 0;
    const int family = nt_v2_family(p, out_f32, &f);
    // measurement kernels kept outside the product (tools/experiments/: gemm_nt_v5, gemm_nt_v8) hook in here when their
    // library is the one loaded (LNX_LIB_PATH=tools/liblnx_experiments.so); the shipped library never sets the hook
    if (g_nt_experiment && g_nt_experiment(p, f, out_f32, st) == 0) return 0;
    if (family == LNX_NT_KERNEL_V7) {
        note_nt_kernel(LNX_NT_KERNEL_V7);
        return launch_nt_v7(p, f, out_f32, st);
    }
    if (family == LNX_NT_KERNEL_V9) {
        note_nt_kernel(LNX_NT_KERNEL_V9);
        return launch_nt_v9(p, f, out_f32, st);
    }
    if (family == LNX_NT_KERNEL_V4) {
        note_nt_kernel(LNX_NT_KERNEL_V4);
        p.tiles_m = cdiv(p.M, BM4);
        p.tiles_n = p.N / BN4;
        const int grid4 = p.tiles_m * p.tiles_n;
        const size_t lds4 = NST4 * STAGE4;
#define V4_LAUNCH(O, FF)                                                                                                             \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v4_kernel<O, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_v4_kernel<O, FF>), dim3(grid4), dim3(512), lds4, st, p);                                         \
    } while (0)
        if (out_f32) V4_LAUNCH(true, F_BIAS | F_RES);
        else if (f == 0) V4_LAUNCH(false, 0);
        else if (f == F_BIAS) V4_LAUNCH(false, F_BIAS);
        else if (f == (F_BIAS | F_C2 | F_GELU)) V4_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
        else if (f == (F_BIAS | F_GELU)) V4_LAUNCH(false, F_BIAS | F_GELU);
        else V4_LAUNCH(false, F_GELU_BWD);
#undef V4_LAUNCH
        return 0;
    }
    note_nt_kernel(LNX_NT_KERNEL_V2);
#define V2_LAUNCH(O, P, FF)                                                                                                          \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v2_kernel<O, P, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_v2_kernel<O, P, FF>), dim3(grid), dim3(512), lds, st, p);                                        \
    } while (0)
    if (f == F_GENERIC) {
        if (out_f32 && patch) V2_LAUNCH(true, true, F_GENERIC);
        else if (out_f32) V2_LAUNCH(true, false, F_GENERIC);
        else if (patch) V2_LAUNCH(false, true, F_GENERIC);
        else V2_LAUNCH(false, false, F_GENERIC);
    } else if (out_f32) {
        V2_LAUNCH(true, false, F_BIAS | F_RES);
    } else if (f == 0) {
        V2_LAUNCH(false, false, 0);
    } else if (f == F_BIAS) {
        V2_LAUNCH(false, false, F_BIAS);
    } else if (f == (F_BIAS | F_C2 | F_GELU)) {
        V2_LAUNCH(false, false, F_BIAS | F_C2 | F_GELU);
    } else if (f == (F_BIAS | F_GELU)) {
        V2_LAUNCH(false, false, F_BIAS | F_GELU);
    } else {
        V2_LAUNCH(false, false, F_GELU_BWD);
    }
#undef V2_LAUNCH
    return 0;
}

// ------------------------------------------------------------------------------------
// gemm_tn v2: dW[N,K] += dY[M,N]^T . A[M,K] with the same LDS-DMA ring (3 stages of 64
// contraction rows, two in flight).  R operand = dY (RW output rows), C operand = A (CW output
// columns), (RW, CW) = (256,128) or (128,256); 8 waves, each a 64x64 output sub-tile.
// Both operands are contracted over their ROW index, so fragments are read transposed with
// ds_read_b64_tr_b16; the 16-byte-chunk XOR key (row & 7) << 1 (applied on the DMA source side)
// makes every 32-lane half of a transposed read hit 16 distinct 16-byte bank slots.
// Requires bf16, M % 64 == 0, plain (non-patch) A.
// ------------------------------------------------------------------------------------
constexpr int TBM = 64;

#define DS_READ_TR2(lo, hi, addr, off2)                                                                                  \
    do {                                                                                                                 \
        const uint32_t a_ = (addr);                                                                                      \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_) : "memory");                                       \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a_), "i"(off2) : "memory");                 \
    } while (0)

template <int RW, int CW, bool PATCH>
__global__ __launch_bounds__(512) void gemm_tn_v2_kernel(const WgradP p) {
    typedef bf16_t T;
    constexpr int RROW = RW * 2, CROW = CW * 2;            // bytes per LDS row of each tile
    constexpr int RBYTES = TBM * RROW, CBYTES = TBM * CROW;
    constexpr int STG = RBYTES + CBYTES;                    // 48 KiB
    constexpr int RINS = RBYTES / 1024, CINS = CBYTES / 1024;
    constexpr int NINS = (RINS + CINS) / 8;                 // wave-instructions per wave per stage = 6
    constexpr int WC = CW / 64;                             // waves along the C operand
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int s = lane & 15, g = lane >> 4;

    // workgroups of one split (the same contraction rows, different output tiles) are consecutive logical ids, i.e.
    // on one XCD at about the same time: its L2 then serves the tiles_k re-reads of dY and the tiles_n re-reads of A
    int bid = xcd_remap(blockIdx.x, p.tiles_n * p.tiles_k * p.splits);
    const int split = bid / (p.tiles_n * p.tiles_k);
    bid -= split * (p.tiles_n * p.tiles_k);
    const int tc = bid % p.tiles_k;
    const int tr = bid / p.tiles_k;
    const int n0 = tr * RW, k0 = tc * CW;
    const int m_begin = split * p.m_per_split;
    const int m_end = min(p.M, m_begin + p.m_per_split);
    if (m_begin >= m_end) return;
    const int ntile = (m_end - m_begin) / TBM;

    const unsigned char* src[NINS];
    int64_t step[NINS];
    // PATCH: the patch origin of contraction row m = (b, ho, wo) of a 2x2 / stride-2 gather from [B, Hin, Win, Cin] is
    // 2 Cin (m + (m / Wo) Wo) elements into the source (Hin = 2 Ho, Win = 2 Wo), so a piece keeps r = m % Wo and
    // po = m + (m / Wo) Wo and advances both by additions when m moves on by a contraction tile -- the closed form
    // (two divisions and two remainders by run-time values per piece and tile) cost ~100 VALU instructions a piece.
    int pr[NINS], po[NINS];
    const int Wo = PATCH ? p.pg.Win >> 1 : 1;
    const int qa = TBM / Wo, qb = TBM - qa * Wo;  // uniform
#pragma unroll
    for (int j = 0; j < NINS; ++j) {
        pr[j] = po[j] = 0;
        const int i = wave + 8 * j;
        if (i < RINS) {
            constexpr int CPR = RW / 8, RPI = 64 / CPR;  // chunks per row, rows per wave-instruction
            const int row = i * RPI + lane / CPR, slot = lane % CPR;
            int col = n0 + ((slot ^ ((row & 7) << 1)) << 3);
            const int lim = ((p.N + 7) >> 3) << 3;
            if (col >= lim) col = lim - 8;
            src[j] = p.dY + ((int64_t)(m_begin + row) * p.lddy + col) * 2;
            step[j] = (int64_t)TBM * p.lddy * 2;
        } else {
            constexpr int CPR = CW / 8, RPI = 64 / CPR;
            const int row = (i - RINS) * RPI + lane / CPR, slot = lane % CPR;
            int col = k0 + ((slot ^ ((row & 7) << 1)) << 3);
            const int lim = ((p.K + 7) >> 3) << 3;
            if (col >= lim) col = lim - 8;
            if (PATCH) {
                // 2x2 patch gather from the NHWC source: row m -> patch origin (recomputed per contraction tile, the row
                // stride is not constant), column k = (kh, kw, c) -> offset inside the patch; a 16-byte chunk never
                // straddles a (kh) segment because 2 * Cin % 8 == 0
                src[j] = p.A + patch_col(p.pg, col) * 2;
                step[j] = (int64_t)p.pg.Cin * 4;  // bytes per unit of po
                const int m = m_begin + row, t = m / Wo;
                pr[j] = m - t * Wo;
                po[j] = m + t * Wo;
            } else {
                src[j] = p.A + ((int64_t)(m_begin + row) * p.lda + col) * 2;
                step[j] = (int64_t)TBM * p.lda * 2;
            }
        }
    }
    auto issue_tile = [&](int stage) {
#pragma unroll
        for (int j = 0; j < NINS; ++j) {
            const int i = wave + 8 * j;
            const unsigned char* gp = src[j];
            if (PATCH && i >= RINS) {
                gp += (int64_t)po[j] * step[j];
                const int r = pr[j] + qb;
                const bool wrap = r >= Wo;
                pr[j] = wrap ? r - Wo : r;
                po[j] += TBM + (qa + (wrap ? 1 : 0)) * Wo;
            } else {
                src[j] += step[j];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(smem + stage * STG + i * 1024), 16, 0, 0);
        }
    };

    // lane-constant fragment offsets (relative to the stage base): transposed block of rows
    // ks*32 + 4g + q (+16), columns 16 i + 4 pp .. +3 of the wave's 64-column slice
    const int q = s >> 2, pp = s & 3;
    uint32_t offR[2][4], offC[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int row = ks * 32 + 4 * g + q;
        const int key = (row & 7) << 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cr = (wr * 64 + i * 16 + 4 * pp) >> 3, cc = (wc * 64 + i * 16 + 4 * pp) >> 3;
            offR[ks][i] = row * RROW + ((cr ^ key) << 4) + (pp & 1) * 8;
            offC[ks][i] = RBYTES + row * CROW + ((cc ^ key) << 4) + (pp & 1) * 8;
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    f32x4_t acc[4][4], accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.db != nullptr && tc == 0 && wc == 0;
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);

    issue_tile(0);
    if (ntile > 1) issue_tile(1);
    // Ping-pong schedule as in gemm_nt_v2_kernel: waves 0-3 / 4-7 run half a contraction tile apart, so the 32
    // transposed fragment reads and the 6 LDS-DMA pieces of one wave are issued under the 40 MFMAs of its SIMD
    // partner (in lock step a tile took ~2600 cycles on an otherwise idle chip against 1280 cycles of MFMA).
    const int grp = wave >> 2;
    if (ntile > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ntile; ++t) {
        const uint32_t st = lds_base + (t % NSTAGE) * STG;
        uint2 rl[2][4], rh[2][4], cl[2][4], chh[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                DS_READ_TR2(rl[ks][i], rh[ks][i], st + offR[ks][i], 16 * RROW);
                DS_READ_TR2(cl[ks][i], chh[ks][i], st + offC[ks][i], 16 * CROW);
            }
        if (t + 2 < ntile) {
            issue_tile((t + 2) % NSTAGE);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 rf[4], cf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rf[i] = make_uint4(rl[ks][i].x, rl[ks][i].y, rh[ks][i].x, rh[ks][i].y);
                cf[i] = make_uint4(cl[ks][i].x, cl[ks][i].y, chh[ks][i].x, chh[ks][i].y);
            }
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) Mfma<T>::run(acc[ri][ci], rf[ri], cf[ci]);
            if (do_bias) {
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) Mfma<T>::run(accb[ri], rf[ri], ones);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp && t + 1 == ntile)) __builtin_amdgcn_s_barrier();
    }
    if (p.ws) {
        // split-K partials: this workgroup's whole RW x CW tile, unmasked, to its slot of the workspace
        // (round 4: in the lanes' own order -- slot ((wave 4 + ri) 4 + ci) 64 + lane holds the float4 acc[ri][ci], i.e. rows 4g .. 4g+3 of one
        // column: one 1-KiB store per instruction instead of four 64-byte row segments; tn_reduce_body reads the same order)
        const int tiles = p.tiles_n * p.tiles_k;
        float4* wt = reinterpret_cast<float4*>(p.ws + (size_t)(split * tiles + tr * p.tiles_k + tc) * (RW * CW)) + (wave * 16) * 64 + lane;
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) wt[(ri * 4 + ci) * 64] = make_float4(acc[ri][ci][0], acc[ri][ci][1], acc[ri][ci][2], acc[ri][ci][3]);
        if (do_bias && s == 0) {
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p.ws[(size_t)p.splits * tiles * (RW * CW) + (size_t)(split * p.tiles_n + tr) * RW + wr * 64 + ri * 16 + 4 * g + r] = accb[ri][r];
        }
        return;
    }
#pragma unroll
    for (int ri = 0; ri < 4; ++ri) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wr * 64 + ri * 16 + 4 * g + r;
            if (n >= p.N) continue;
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int k = k0 + wc * 64 + ci * 16 + s;
                if (k >= p.k_store) continue;
                int col = k;
                if (p.k_perm_c > 0) {
                    const int P = p.K / p.k_perm_c;
                    const int pq = k / p.k_perm_c;
                    col = (k - pq * p.k_perm_c) * P + pq;
                }
                atomicAdd(p.dW + (int64_t)n * p.lddw + col, acc[ri][ci][r]);
            }
            if (do_bias && s == 0) atomicAdd(p.db + n, accb[ri][r]);
        }
    }
}

// second stage of the workspace path: dW[n, col(k)] += sum over splits of the partial tiles, db likewise
__device__ __forceinline__ void tn_reduce_body(const WgradP& p, int RW, int CW, const int64_t idx) {
    const int tiles = p.tiles_n * p.tiles_k;
    const int tile_f4 = RW * CW / 4;  // float4 slots of a partial tile, in the producing lanes' order (gemm_tn_v2_kernel's epilogue)
    const size_t tile_floats = (size_t)RW * CW;
    if (idx < (int64_t)tiles * tile_f4) {
        const int tile = (int)(idx / tile_f4), f = (int)(idx - (int64_t)tile * tile_f4);
        const int tr = tile / p.tiles_k, tc = tile - tr * p.tiles_k;
        const int lane = f & 63, ci = (f >> 6) & 3, ri = (f >> 8) & 3, wave = f >> 10;
        const int WC = CW / 64, wr = wave / WC, wc = wave - wr * WC;
        const int n0 = tr * RW + wr * 64 + ri * 16 + 4 * (lane >> 4);  // rows n0 .. n0 + 3
        const int k = tc * CW + wc * 64 + ci * 16 + (lane & 15);
        if (n0 >= p.N || k >= p.k_store) return;
        const float4* src = reinterpret_cast<const float4*>(p.ws + (size_t)tile * tile_floats) + f;
        // eight independent loads in flight per thread; the summation order is fixed (partials of splits sp = j mod 8
        // are added in order, the remainder goes to sum 0, then the eight sums are combined pairwise)
        float4 part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t sstep = (size_t)tiles * tile_floats / 4;  // float4 units between the same tile of consecutive splits
        int sp = 0;
        for (; sp + 8 <= p.splits; sp += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(sp + j) * sstep];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                part[j].x += v[j].x;
                part[j].y += v[j].y;
                part[j].z += v[j].z;
                part[j].w += v[j].w;
            }
        }
        for (; sp < p.splits; ++sp) {
            const float4 v = src[(size_t)sp * sstep];
            part[0].x += v.x;
            part[0].y += v.y;
            part[0].z += v.z;
            part[0].w += v.w;
        }
#pragma unroll
        for (int w = 4; w > 0; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) {
                part[j].x += part[j + w].x;
                part[j].y += part[j + w].y;
                part[j].z += part[j + w].z;
                part[j].w += part[j + w].w;
            }
        const float av[4] = {part[0].x, part[0].y, part[0].z, part[0].w};
        int col = k;
        if (p.k_perm_c > 0) {
            const int P = p.K / p.k_perm_c;
            const int pq = k / p.k_perm_c;
            col = (k - pq * p.k_perm_c) * P + pq;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + r < p.N) p.dW[(int64_t)(n0 + r) * p.lddw + col] += av[r];  // (16 lanes = 16 consecutive k: 64-byte segments per row)
    } else if (p.db != nullptr) {
        const int64_t n = idx - (int64_t)tiles * tile_f4;
        if (n < p.N) {
            const int tr = (int)n / RW;
            const float* src = p.ws + (size_t)p.splits * tiles * tile_floats + (size_t)tr * RW + (n - tr * RW);
            float a = 0.f;
            for (int sp = 0; sp < p.splits; ++sp) a += src[(size_t)sp * p.tiles_n * RW];
            p.db[n] += a;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const WgradP p, int RW, int CW) {
    tn_reduce_body(p, RW, CW, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Round 4: the second stages of several weight-gradient products in ONE launch (lnx_wgrad_args.defer / lnx_gemm_tn_flush): the four
// products of a RoPE block or the two of a ConvNeXt block leave their partial tiles in separate workspace regions and one kernel sums
// them all -- 37 launches of ~11 us per step become 13, each with several times the parallelism of a single reduce.
constexpr int TN_BATCH = 8;
struct TnReduceDesc {
    WgradP p;
    int RW, CW, block_start;
};
struct TnReduceBatch {
    TnReduceDesc d[TN_BATCH];
    int n;
};
__global__ __launch_bounds__(256) void gemm_tn_reduce_batch_kernel(const TnReduceBatch b) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < TN_BATCH; ++i)
        if (i < b.n && (int)blockIdx.x >= b.d[i].block_start) j = i;
    tn_reduce_body(b.d[j].p, b.d[j].RW, b.d[j].CW, (int64_t)((int)blockIdx.x - b.d[j].block_start) * 256 + threadIdx.x);
}

// deferred second stages of this host thread (one training loop = one thread = one stream; a product deferred on another stream
// flushes what is pending first)
static thread_local TnReduceBatch g_tn_pending = {};
static thread_local int g_tn_pending_blocks = 0;
static thread_local hipStream_t g_tn_pending_stream = nullptr;

// The postponed second stages run on the stream their products were launched on; `st` must be that stream (a flush asked for on another
// stream would order the reduces behind the wrong work) -- nullptr stands for "whichever it was" (the discard / cleanup paths).
int tn_flush(hipStream_t st) {
    if (g_tn_pending.n == 0) return 0;
    if (st != nullptr && st != g_tn_pending_stream) return 1;
    hipLaunchKernelGGL(gemm_tn_reduce_batch_kernel, dim3((unsigned)g_tn_pending_blocks), dim3(256), 0, g_tn_pending_stream, g_tn_pending);
    g_tn_pending.n = 0;
    g_tn_pending_blocks = 0;
    return 0;
}

// Forget the postponed second stages of this host thread without running them (error paths, plan teardown: their descriptors hold raw
// workspace / gradient pointers that may not outlive the call that failed).  Returns how many were dropped.
int tn_discard() {
    const int n = g_tn_pending.n;
    g_tn_pending.n = 0;
    g_tn_pending_blocks = 0;
    g_tn_pending_stream = nullptr;
    return n;
}

bool tn_v2_ok(const WgradP& p, int dtype) {
    return dtype == LNX_BF16 && p.M % TBM == 0 && p.M >= 4096 && p.N >= 8 && p.K >= 8;
}

int launch_tn_v2(const WgradP& p0, int splits_hint, hipStream_t st, bool defer) {
    WgradP p = p0;
    // tile orientation with the least padding waste
    auto waste = [&](int rw, int cw) { return (double)cdiv(p.N, rw) * rw * cdiv(p.K, cw) * cw; };
    static const char* force = getenv("LNX_TN_ORIENT");  // A/B switch: "r" = 256x128 tiles, "c" = 128x256
    const bool wide_r = force ? force[0] == 'r' : waste(256, 128) <= waste(128, 256);
    const int RW = wide_r ? 256 : 128, CW = wide_r ? 128 : 256;
    p.tiles_n = cdiv(p.N, RW);
    p.tiles_k = cdiv(p.K, CW);
    const int tiles = p.tiles_n * p.tiles_k;
    const int mtiles = p.M / TBM;
    int splits = splits_hint;
    if (splits <= 0) {
        splits = 256 / tiles;  // the 144 KiB ring allows one workgroup per CU: at most one full round
        if (splits < 1) splits = 1;
        const int max_splits = mtiles / 8 > 0 ? mtiles / 8 : 1;  // ... with >= 8 contraction tiles each
        if (splits > max_splits) splits = max_splits;
    }
    if (splits > mtiles) splits = mtiles;
    if (splits < 1) splits = 1;
    const int per = cdiv(mtiles, splits);
    p.m_per_split = per * TBM;
    p.splits = cdiv(mtiles, per);
    const int grid = tiles * p.splits;
    const size_t lds = NSTAGE * (size_t)(TBM * (RW + CW) * 2);
#define TNV2_(R, C, P)                                                                                                               \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_v2_kernel<R, C, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_tn_v2_kernel<R, C, P>), dim3(grid), dim3(512), lds, st, p);                                         \
    } while (0)
#define TNV2(R, C)                                           \
    do {                                                     \
        if (p.a_mode == LNX_ADDR_PATCH2) TNV2_(R, C, true);  \
        else TNV2_(R, C, false);                             \
    } while (0)
    static const bool no_ws = getenv("LNX_TN_ATOMIC") != nullptr;  // A/B switch for benchmarking
    const size_t need = (size_t)p.splits * tiles * (256 * 128) + (size_t)p.splits * p.tiles_n * RW;
    // padded tiles are stored and re-read whole: with poorly filled tiles (N x K well below tiles x 256 x 128) the atomics move fewer bytes
    const bool sparse_tiles = (double)p.N * p.k_store < 0.7 * (double)tiles * (256 * 128);
    if (no_ws || p.ws == nullptr || (size_t)p.ws_floats < need || p.splits < 2 || sparse_tiles) p.ws = nullptr;
    if (wide_r) TNV2(256, 128);
    else TNV2(128, 256);
#undef TNV2
#undef TNV2_
    if (p.ws) {
        const int64_t work = (int64_t)tiles * (RW * CW / 4) + (p.db ? p.N : 0);  // one thread per float4 slot of a partial tile, then one per bias entry
        const int blocks = (int)cdiv(work, 256);
        if (defer) {
            if (g_tn_pending.n > 0 && (g_tn_pending_stream != st || g_tn_pending.n == TN_BATCH)) (void)tn_flush(nullptr);
            TnReduceDesc& d = g_tn_pending.d[g_tn_pending.n++];
            d.p = p;
            d.RW = RW;
            d.CW = CW;
            d.block_start = g_tn_pending_blocks;
            g_tn_pending_blocks += blocks;
            g_tn_pending_stream = st;
        } else {
            hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, RW, CW);
        }
    }
    return 0;
}

}  // namespace lnxg
