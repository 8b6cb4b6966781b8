// gemm_nt v9 (round 4): the 256x256 / BK = 32 / four-stage ring / ping-pong K loop of gemm_nt_v4 as a PERSISTENT kernel that draws
// its tiles from the per-XCD atomic counters (common.hpp) and never lets the ring drain.
//
// Why.  One-shot gemm_nt_v4 spends a quarter of a tile's life outside the K loop at K = 1024: the first three slices'
// round trip at the start (one workgroup per CU: nothing else runs meanwhile) and the epilogue at the end.  Here the last three K
// iterations of a tile fetch the NEXT tile's first three slices, so the prologue latency disappears under the epilogue, and the
// epilogue's stores drain under the first two iterations of the next tile: stores, LDS-DMA and loads share one in-order counter,
// so those two iterations wait with the stores' count added (a compile-time number for a full tile; a partial last row tile
// drains instead), i.e. for "slice k+1 landed" and not for "every store acknowledged".  What v9 does NOT do is gemm_nt_v7's
// deferral of the stores into the next K loop: a 128x64 wave tile's results are 64-128 registers on top of 128 accumulator registers,
// which the 256-register budget of two waves per SIMD does not hold (v7's 64x64 wave tiles do).
//
// Scheduling: as gemm_nt_v7 -- wave 0 draws the position after next in iteration 0 (one more operation on ITS counter), publishes it
// through an LDS dword in iteration 2, everyone reads it in iteration 3; first use in iteration nk - 3 >= 5.  K / 32 >= 8.
#include "gemm_common.hpp"

namespace lnxg {

#define V9_READ4(dst, addr)                                                                             \
    do {                                                                                                \
        const uint32_t a_ = (addr);                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                        \
        asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst[1]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

template <int N> struct IC9 { static constexpr int value = N; };

constexpr int BM9 = 256, BN9 = 256, BK9 = 32, ROWB9 = 64;
constexpr int STAGE9 = (BM9 + BN9) * ROWB9;  // 32 KiB
constexpr int NST9 = 4;
constexpr int PIECES9 = STAGE9 / 1024 / 8;   // 1-KiB LDS-DMA instructions per wave and slice = 4

__device__ __forceinline__ int key9(int row) { return (row & 16) ? 3 : 0; }

// 16-byte stores one lane issues in the epilogue of a FULL tile (two 64x64 sub-tiles): what the first two iterations of the next
// tile let stay in flight.  Checked against the compiled code by tools/audit_counted_waits.py
// (run by __graft_entry__.build() and tests/test_host_logic.py: a mismatch fails the build; a smaller number is always safe).
template <bool OUT_F32, int F> struct EpiStores {
    static constexpr int value = 2 * (OUT_F32 ? 16 : 8) + ((F & F_C2) ? 16 : 0);
};

template <bool OUT_F32, int F>
__global__ __launch_bounds__(512) void gemm_nt_v9_kernel(const GemmP p) {
    typedef bf16_t T;
    constexpr int NSTORE = EpiStores<OUT_F32, F>::value;
    static_assert(2 * PIECES9 + NSTORE + 1 <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST9][A 256 rows | W 256 rows][64 B] + one dword

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // waves 0-3 / 4-7 (the ping-pong groups) cover the column halves
    const int s = lane & 15, g = lane >> 4;
    const int grp = wave >> 2;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK9;  // >= 8

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const uint32_t chunk_off = (uint32_t)((g ^ (((s >> 2) & 1) * 3)) << 4);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = (uint32_t)((wm * 128 + frag_row) * ROWB9) + chunk_off;
    const uint32_t w_off = (uint32_t)((BM9 + wn * 64 + frag_row) * ROWB9) + chunk_off;
    const uint32_t slot_addr = lds_base + NST9 * STAGE9;

    // LDS-DMA sources as 32-bit byte offsets from p.A / p.W (nt_v9_ok checks the range), of this tile and of the next one
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t src = {0, 0, 0, 0}, srcn = {0, 0, 0, 0};
    auto piece_offset = [&](int j, int m0, int n0) __attribute__((always_inline)) -> uint32_t {
        const int i = wave + 8 * j;
        const int row = 16 * i + (lane >> 2);
        const int slot = lane & 3;
        if (row < BM9) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            return (uint32_t)(m * (int)p.lda + (slot ^ key9(row)) * 8) * 2u;
        }
        const int wr = row - BM9;
        return (uint32_t)((n0 + wr) * (int)p.ldw + (slot ^ key9(wr)) * 8) * 2u;  // N % 256 == 0: every W row exists
    };
    auto issue = [&](bool of_next, int kslice, int stage) __attribute__((always_inline)) {
        const u32x4_t sv = of_next ? srcn : src;
#pragma unroll
        for (int j = 0; j < PIECES9; ++j) {
            const int i = wave + 8 * j;
            const unsigned char* base = 16 * i < BM9 ? p.A : p.W;  // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (sv[j] + (uint32_t)(kslice * BK9 * 2))),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE9 + i * 1024), 16, 0, 0);
        }
    };

    // ---- this workgroup's share of the tiles and its first draw (see gemm_nt_v7) ----
    const TileShare sh = tile_share(ntiles);
    const int xcnt = sh.cnt, xbase = sh.base, xgrid = sh.workers;
    const bool dyn = p.tile_slot >= 0;
    unsigned* const ctr = &g_tile_ctr[dyn ? p.tile_slot : 0][sh.part][0];
    bool reset_ctr = false;
    int pos = sh.index, m0, n0;
    if (dyn) {
        if (wave == 0) {
            const uint32_t one = 1u;
            uint32_t first;
            uint64_t save;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_waitcnt vmcnt(0)\n\tds_write_b32 %4, %0\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(first), "=&s"(save) : "v"(ctr), "v"(one), "v"(slot_addr) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        uint32_t seen0;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen0) : "v"(slot_addr) : "memory");
        pos = __builtin_amdgcn_readfirstlane((int)seen0);
        reset_ctr = pos == xcnt + xgrid - 1;
        __builtin_amdgcn_s_barrier();
    }
    if (pos >= xcnt) {
        if (reset_ctr && tid == 0) sched_reset(ctr);
        return;
    }
    auto pos_origin = [&](int ps, int& mo, int& no) __attribute__((always_inline)) {
        const int logical = xbase + ps;
        no = (logical % p.tiles_n) * BN9;
        mo = (logical / p.tiles_n) * BM9;
    };
    pos_origin(pos, m0, n0);
#pragma unroll
    for (int j = 0; j < PIECES9; ++j) src[j] = piece_offset(j, m0, n0);
    issue(false, 0, 0);
    issue(false, 1, 1);
    issue(false, 2, 2);
    int ring = 0;  // LDS stage of the current K slice; runs on across tiles
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES9) : "memory");  // slice 0 landed
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();  // ping-pong: waves 4-7 run one barrier interval behind waves 0-3
    bool stores_behind = false;  // the previous tile's epilogue stores (exactly NSTORE per lane) are the youngest operations before this tile

    while (true) {
        int next = pos + xgrid;
        bool has_next = !dyn && next < xcnt;
        int nm0 = 0, nn0 = 0;
        uint32_t fetched = 0, seen = 0;
        f32x4_t acc0[4][4], acc1[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        // One K iteration.  EX = how many of the previous tile's stores may still be in flight behind the slice this iteration waits
        // for (NSTORE in iterations 0 and 1 of a tile that follows a full tile, else 0); FX: 1 draw, 2 publish, 3 read (as v7).
        auto kstep = [&](int kt, auto EX, auto FX) __attribute__((always_inline)) {
            constexpr int ex = decltype(EX)::value, fx = decltype(FX)::value;
            const uint32_t stg = lds_base + ring * STAGE9;
            uint4 wf[4], af0[4], af1[4];
            V9_READ4(wf, stg + w_off);
            V9_READ4(af0, stg + a_off);
            V9_READ4(af1, stg + a_off + 64 * ROWB9);
            if (fx == 3 && dyn) asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(slot_addr) : "memory");
            const bool draw = fx == 1 && dyn && wave == 0;
            if (draw) sched_draw(fetched, ctr);
            asm volatile("" ::: "memory");
            const int stage3 = (ring + 3) & 3;
            // 0: nothing issued, 1: issued.  Slice kt + 3 of this tile, or slice kt + 3 - nk of the next one
            int issued = 1;
            if (kt + 3 < nk) issue(false, kt + 3, stage3);
            else if (has_next) issue(true, kt + 3 - nk, stage3);
            else issued = 0;
            const bool ext = ex > 0 && stores_behind;  // wave-uniform
#define V9_WAIT(n)                                                                   \
    do {                                                                             \
        if (draw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) + 1) : "memory");     \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");                \
    } while (0)
            if (issued) {
                // slices kt + 2 and kt + 3 (2 x 4 pieces) stay in flight; in iterations 0 / 1 after a full tile the stores sit between
                // slice kt + 2 (issued before them) and slice kt + 3 (just issued): they may stay too
                if (ext) V9_WAIT(2 * PIECES9 + ex);
                else V9_WAIT(2 * PIECES9);
            } else {
                // the last tile's last three iterations: slices kt + 1, kt + 2 may be all that is left -- keep it simple
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES9) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#undef V9_WAIT
            if (fx == 2 && dyn && wave == 0) {
                uint64_t save;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(save) : "v"(slot_addr), "v"(fetched) : "memory");
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
            ring = (ring + 1) & 3;
        };
        // (the barrier that closes an iteration is outside kstep: the very last one of a workgroup is skipped by waves 4-7, which made
        // one more at the start)
        // The draw of iteration 0 is YOUNGER than the previous tile's stores, which iterations 0 and 1 let stay in flight: only iteration
        // 2's wait (everything but the two youngest slices) retires it -> publish there, read in iteration 3.
        kstep(0, IC9<NSTORE>(), IC9<1>());
        __builtin_amdgcn_s_barrier();
        kstep(1, IC9<NSTORE>(), IC9<0>());
        __builtin_amdgcn_s_barrier();
        kstep(2, IC9<0>(), IC9<2>());
        __builtin_amdgcn_s_barrier();
        kstep(3, IC9<0>(), IC9<3>());
        __builtin_amdgcn_s_barrier();
        if (dyn) {
            next = __builtin_amdgcn_readfirstlane((int)seen);
            has_next = next < xcnt;
            reset_ctr = reset_ctr || next == xcnt + xgrid - 1;
        }
        if (has_next) {
            pos_origin(next, nm0, nn0);
#pragma unroll
            for (int j = 0; j < PIECES9; ++j) srcn[j] = piece_offset(j, nm0, nn0);
        }
        for (int kt = 4; kt < nk; ++kt) {
            kstep(kt, IC9<0>(), IC9<0>());
            if (!(grp && kt + 1 == nk && !has_next)) __builtin_amdgcn_s_barrier();
        }
        // ---- epilogue: the one-shot kernels' (its loads wait for the counter to drain, i.e. also for the next tile's three slices --
        // which the next K loop needs at once anyway); its stores stay in flight into the next tile ----
        gemm_epilogue_fast<T, OUT_F32, F>(p, acc0, m0 + wm * 128, n0 + wn * 64, lane);
        gemm_epilogue_fast<T, OUT_F32, F>(p, acc1, m0 + wm * 128 + 64, n0 + wn * 64, lane);
        if (!has_next) break;
        const bool full = m0 + BM9 <= p.M;
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows beyond M: fewer stores than NSTORE were issued, count nothing
        stores_behind = full;
        asm volatile("" ::: "memory");
        pos = next;
        m0 = nm0;
        n0 = nn0;
        src = srcn;
    }
    if (reset_ctr && tid == 0) sched_reset(ctr);
}

bool nt_v9_ok(const GemmP& p, int f, bool out_f32) {
    if (f == (int)F_GENERIC || p.a_mode == LNX_ADDR_PATCH2) return false;
    if (p.N % BN9 != 0 || p.K % BK9 != 0 || p.K / BK9 < 8) return false;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)p.M * p.lda * 2 >= lim || (int64_t)p.N * p.ldw * 2 >= lim) return false;
    if (out_f32) return f == (F_BIAS | F_RES);
    return f == 0 || f == F_BIAS || f == (F_BIAS | F_C2 | F_GELU) || f == (F_BIAS | F_GELU) || f == F_GELU_BWD;
}

int launch_nt_v9(const GemmP& p0, int f, bool out_f32, hipStream_t st) {
    GemmP p = p0;
    p.tiles_m = cdiv(p.M, BM9);
    p.tiles_n = p.N / BN9;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int cus = device_cus();
    if (cus <= 0) return 1;
    const int room = persistent_cus(cus);
    const int grid = ntiles < room ? ntiles : room;
    const size_t lds = NST9 * STAGE9 + 16;
    p.tile_slot = tile_sched_static() ? -1 : tile_slot_of(st);
#define V9_LAUNCH(O, FF)                                                                                                             \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v9_kernel<O, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_v9_kernel<O, FF>), dim3(grid), dim3(512), lds, st, p);                                           \
    } while (0)
    if (out_f32) V9_LAUNCH(true, F_BIAS | F_RES);
    else if (f == 0) V9_LAUNCH(false, 0);
    else if (f == F_BIAS) V9_LAUNCH(false, F_BIAS);
    else if (f == (F_BIAS | F_C2 | F_GELU)) V9_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
    else if (f == (F_BIAS | F_GELU)) V9_LAUNCH(false, F_BIAS | F_GELU);
    else V9_LAUNCH(false, F_GELU_BWD);
#undef V9_LAUNCH
    return 0;
}

}  // namespace lnxg
