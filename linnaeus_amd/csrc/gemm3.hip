// gemm_nt v7 (round 3): the 256x128 / 8-wave / ping-pong K loop of gemm_nt_v2 as a PERSISTENT kernel whose epilogue
// traffic rides under the next tile's K loop.
//
// Why.  In the model the forward / data-gradient GEMMs of the RoPE blocks store (and load) as many bytes as their K loop
// fetches: fc1 writes two [M, hidden] tensors, the GELU' data gradient reads one and writes one, proj / fc2 read and write the
// fp32 residual stream.  With one workgroup per CU and every workgroup starting together, the whole chip is in the K loop
// (HBM idle), then the whole chip is in the epilogue (matrix cores idle, HBM saturated): the two phases ADD
// (profiles/r03_nt_v5_phase_timeline.log; a second co-resident workgroup does not change it, the phases re-align within two
// rounds because the saturated epilogue phase is a shared queue).  Here the overlap is built into the instruction stream:
//
//   tile t:  K loop | epilogue ARITHMETIC only: results stay in registers (32..64 per lane)
//   tile t+1: the first HEAD K iterations each issue a third of tile t's stores in their memory phase;
//             TAIL iterations before the last one fetch tile t+1's own epilogue operands (GELU' input / fp32 residual);
//             the next tile's first two K slices are issued BEFORE the epilogue arithmetic (prologue latency hidden too).
//
// Stores, operand fetches and LDS-DMA share one in-order counter, so every extra operation is issued BEFORE its
// iteration's LDS-DMA pieces: the wait for slice k+1 (issued one iteration earlier) then has to skip exactly this
// iteration's extras plus the six pieces of slice k+2 -- a compile-time count, because the HEAD / TAIL iterations are
// peeled (straight-line code: no load sits under a branch).  K / 64 >= 6.
//
// Round 4: tiles come from per-XCD ATOMIC COUNTERS instead of `tile += gridDim.x`.  A persistent workgroup owns its CU's LDS and
// registers; beside an RCCL kernel (data-parallel training: the gradient all-reduce of the previous backward segment is resident on
// 32-64 CUs for hundreds of microseconds) the workgroups that find no free CU start only when a sibling has finished -- with a static
// stride each of them still owes its whole share, the launch takes two rounds instead of (256 / free CUs) of one.  With a counter
// the early workgroups eat the tiles and a latecomer finds none.  The XCD-local tile order is kept: XCD x owns the contiguous
// logical range xcd_remap() gives it and a counter of its own (workgroup b runs on XCD b & 7 under round-robin dispatch); nothing
// is stolen across XCDs (a collective's workgroups are dealt round-robin too).  Mechanics: wave 0 fetches the tile AFTER next with
// one lane (global_atomic_add with return, issued in iteration 0 in front of the LDS-DMA pieces and counted like any other extra of
// the in-order counter), publishes it through one LDS dword in iteration 1, every wave reads it in iteration 2 -- two iterations
// before the first use (the ring refill at nk - 2).  The fetch that returns the range's last index is the last one the counter
// will see in this launch: that workgroup zeroes it for the next launch.  LNX_TILE_SCHED=static restores the stride (A/B).
#include "gemm_common.hpp"

namespace lnxg {

#define V7_READ4(dst, addr)                                                                              \
    do {                                                                                                 \
        const uint32_t a_ = (addr);                                                                      \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                         \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[1]) : "v"(a_) : "memory");              \
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:1536" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)

template <int N> struct IC { static constexpr int value = N; };

constexpr int BM7 = 256, BN7 = 128, BK7 = 64;
constexpr int STAGE7 = (BM7 + BN7) * ROWB;  // 48 KiB
constexpr int NST7 = 3;
constexpr int PIECES7 = (BM7 + BN7) / 8 / 8;  // 6 LDS-DMA wave-instructions (1 KiB each) per wave and K slice

// inline-asm global fetches: the compiler must not track them (beside LDS-DMA in flight it would wait vmcnt(0) at their first
// use, i.e. for the slices prefetched for the NEXT tile); the counted waits of the K loop cover them
__device__ __forceinline__ void asm_load16(uint4& d, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void asm_load4(float& d, const void* ptr) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(ptr) : "memory");
}

// HEAD7 / TAIL7: K iterations at the start / before the last one of a tile that carry the deferred stores / the operand fetches
template <bool OUT_F32, int F, int HEAD7, int TAIL7>
__global__ __launch_bounds__(512) void gemm_nt_v7_kernel(const GemmP p) {
    typedef bf16_t T;
    constexpr bool TWO = (F & F_C2) != 0;
    constexpr int NSTORE = OUT_F32 ? 16 : (TWO ? 16 : 8);                      // 16-byte stores per lane and tile
    constexpr int NLOAD = (F & F_GELU_BWD) ? 8 : ((F & F_RES) ? 16 : 0);       // 16-byte epilogue-operand fetches per lane and tile
    constexpr int NCONST = ((F & F_BIAS) ? 4 : 0) + ((F & F_RES) ? 4 : 0);     // bias (4 x 16 bytes) / DropPath scale (4 dwords) fetches
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int s = lane & 15, g = lane >> 4;
    const int grp = wave >> 2;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / BK7;  // >= HEAD7 + TAIL7 + 1 (launch_nt_v7 picks HEAD7 / TAIL7 accordingly)

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const int frag_key = ((s >> 1) & 1) | ((s >> 2) << 1);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t ch0 = (uint32_t)((g ^ frag_key) << 4);  // the second 64-byte half of a row is chunk ^ 4: offset ^ 64
    const uint32_t a_off0 = (wm * 64 + frag_row) * ROWB + ch0;
    const uint32_t w_off0 = (BM7 + wn * 64 + frag_row) * ROWB + ch0;

    // LDS-DMA sources as 32-bit byte offsets from p.A / p.W (checked by nt_v7_ok): of the current tile and of the next one,
    // whose first two slices are issued by the last two iterations of the current tile -- the ring never drains
    typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));  // a register vector (an array selected at run time would go to scratch)
    u32x8_t src = {0, 0, 0, 0, 0, 0, 0, 0}, srcn = {0, 0, 0, 0, 0, 0, 0, 0};
    auto piece_offset = [&](int j, int m0, int n0) __attribute__((always_inline)) -> uint32_t {
        const int i = wave + 8 * j;
        const int row = 8 * i + (lane >> 3);
        const int slot = lane & 7;
        if (row < BM7) {
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            return (uint32_t)(m * (int)p.lda + (slot ^ row_key(row)) * 8) * 2u;
        }
        const int wr = row - BM7;
        int n = n0 + wr;
        if (n >= p.N) n = p.N - 1;
        return (uint32_t)(n * (int)p.ldw + (slot ^ row_key(wr)) * 8) * 2u;
    };
    // (nothing here may end up in scratch memory: a scratch access inside the K loop is a vector-memory load that the in-order
    // counter waits for -- check ScratchSize = 0 in -Rpass-analysis=kernel-resource-usage after every change)
    auto issue = [&](bool of_next, int kslice, int stage) __attribute__((always_inline)) {
        const u32x8_t sv = of_next ? srcn : src;
#pragma unroll
        for (int j = 0; j < PIECES7; ++j) {
            const int i = wave + 8 * j;
            const unsigned char* base = 8 * i < BM7 ? p.A : p.W;  // wave-uniform: a piece is all A rows or all W rows
            const uint32_t so = sv[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (so + (uint32_t)(kslice * BK7 * 2))),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE7 + i * 1024), 16, 0, 0);
        }
    };

    // registers that outlive a tile: its results (stored during the next tile's K loop) and where they go.
    // F_C2 | F_GELU (fc1) keeps ONE tensor, the bf16 pre-activation, and writes both outputs from it: C2 = it, C = GELU(it) --
    // the activation of the ROUNDED pre-activation, which is the value the backward differentiates at (GELU'(C2)); the other
    // kernels round GELU(fp32 pre-activation), a difference below bf16 resolution of the result
    uint4 hc[4][2];              // bf16 output (or pre-activation): row slot mi, column halves
    float4 hf[4][4];             // fp32 output
    int pend_row = -1, pend_col = 0;  // first row (slot 0) / first column of this lane's pending sub-tile; -1 = nothing pending
    (void)hf;

    auto store_slot = [&](int slot, bool check) __attribute__((always_inline)) {  // slot is a compile-time constant at every call site
        if (OUT_F32) {
            const int mi = slot >> 2, q = slot & 3, m = pend_row + 4 * mi;
            if (!check || m < p.M) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + ((int64_t)m * p.ldc + pend_col) + 4 * q) = hf[mi][q];
        } else {
            const int sl = TWO ? (slot >> 1) : slot;  // TWO: slots 2k / 2k+1 = C2 / C of the same 16 bytes
            const int mi = sl >> 1, h = sl & 1, m = pend_row + 4 * mi;
            if (TWO && (slot & 1) == 0) {
                if (!check || m < p.M) st16(reinterpret_cast<T*>(p.C2) + ((int64_t)m * p.ldc2 + pend_col) + 8 * h, hc[mi][h]);
            } else if (TWO) {
                Vec16<T> t, o;
                t.raw = hc[mi][h];
                f32x2_t x[4], a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = f32x2_t{t.get(2 * i), t.get(2 * i + 1)};
                gelu_lean2_n<4>(x, a);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o.set(2 * i, a[i].x);
                    o.set(2 * i + 1, a[i].y);
                }
                if (!check || m < p.M) st16(reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + pend_col) + 8 * h, o.raw);
            } else {
                if (!check || m < p.M) st16(reinterpret_cast<T*>(p.C) + ((int64_t)m * p.ldc + pend_col) + 8 * h, hc[mi][h]);
            }
        }
    };

    // this workgroup's XCD range of logical tiles [xbase, xbase + xcnt) and its position `pos` in it (xcd_remap's arithmetic)
    const TileShare sh = tile_share(ntiles);
    const int xcnt = sh.cnt, xbase = sh.base, xgrid = sh.workers;  // (xgrid: workgroups of this launch on this XCD)
    const bool dyn = p.tile_slot >= 0;
    unsigned* const ctr = &g_tile_ctr[dyn ? p.tile_slot : 0][sh.part][0];
    const uint32_t slot_addr = lds_base + NST7 * STAGE7;  // the LDS dword the fetched position travels through
    bool reset_ctr = false;  // this workgroup drew the range's last fetch: it zeroes the counter on its way out
    int pos = sh.index, m0, n0;
    if (dyn) {
        // EVERY tile is drawn, the first one too (0.6 us of L2 round trip per launch): a workgroup that starts late -- its CU was
        // held by a collective kernel -- then owes nothing.  Draws per XCD and launch: one per tile + one failed draw per workgroup,
        // so the draw that returns xcnt + xgrid - 1 is the last the counter sees.
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
        __builtin_amdgcn_s_barrier();  // everyone has read the slot before iteration 1 of the first tile may overwrite it
    }
    if (pos >= xcnt) {
        if (reset_ctr && tid == 0) sched_reset(ctr);
        return;
    }
    auto pos_origin = [&](int ps, int& mo, int& no) __attribute__((always_inline)) {
        const int logical = xbase + ps;
        no = (logical % p.tiles_n) * BN7;
        mo = (logical / p.tiles_n) * BM7;
    };
    pos_origin(pos, m0, n0);
#pragma unroll
    for (int j = 0; j < PIECES7; ++j) src[j] = piece_offset(j, m0, n0);
    issue(false, 0, 0);
    issue(false, 1, 1);
    int ring = 0;  // LDS stage of the current K slice; runs on across tiles
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES7) : "memory");  // slice 0 landed (slice 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();  // ping-pong: waves 4-7 run one barrier interval behind waves 0-3

    while (true) {
        const int mrow0 = m0 + wm * 64, ncol0 = n0 + wn * 64;
        const int nb = ncol0 + g * 16;                      // first of this lane's 16 columns
        const int mbase = mrow0 + (s >> 2) * 16 + (s & 3);  // row of slot mi = mbase + 4 mi
        const bool live = ncol0 < p.N;                      // column tiles beyond N (wave-uniform: N % 64 == 0): nothing to fetch or store
        const bool full = m0 + BM7 <= p.M;                  // only full tiles defer their stores (every lane then issues every store)
        const int nbc = live ? nb : 0;
        // static stride: known now.  Counter: known after iteration 2 (has_next is first looked at in iteration nk - 2 >= 4)
        int next = pos + xgrid;
        bool has_next = !dyn && next < xcnt;
        int nm0 = 0, nn0 = 0;
        uint32_t fetched = 0, seen = 0;  // wave 0 lane 0: the counter's answer; every lane: the published value

        // epilogue operands, fetched in the TAIL iterations: GELU' input (bf16) or fp32 residual, then (last group) the bias of
        // this lane's 16 columns and the DropPath scale of its 4 rows
        uint4 pa[4][2];
        uint4 pr[4][4];  // fp32 residual, as raw 16-byte words
        uint4 bias4[4];
        float rs[4];
        (void)pa;
        (void)pr;
        (void)bias4;
        (void)rs;
        auto load_slot = [&](int slot) __attribute__((always_inline)) {
            if (slot >= NLOAD) {  // constants
                const int c = slot - NLOAD;
                if ((F & F_BIAS) && c < 4) {
                    asm_load16(bias4[c], p.bias + nbc + 4 * c);
                } else {
                    const int mi = (F & F_BIAS) ? c - 4 : c;
                    const bool scaled = p.rowscale != nullptr;
                    const float* rsp = scaled ? p.rowscale : p.bias;  // always a fetch of something valid
                    const int m = min(mbase + 4 * mi, p.M - 1);
                    asm_load4(rs[mi], rsp + (scaled ? m / p.rows_per_sample : 0));
                }
            } else if (F & F_GELU_BWD) {
                const int mi = slot >> 1, h = slot & 1, m = min(mbase + 4 * mi, p.M - 1);
                asm_load16(pa[mi][h], reinterpret_cast<const T*>(p.aux) + ((int64_t)m * p.ldaux + nbc) + 8 * h);
            } else if (F & F_RES) {
                const int mi = slot >> 2, q = slot & 3, m = min(mbase + 4 * mi, p.M - 1);
                asm_load16(pr[mi][q], p.res + ((int64_t)m * p.ldres + nbc) + 4 * q);
            }
        };

        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        const bool pending = __builtin_amdgcn_readfirstlane((int)(pend_row >= 0)) != 0;  // the same for every lane of a wave

        // One K iteration.  ST / LD = index of the store / fetch group issued in its memory phase (-1: none).  Slice kt+2 (of
        // the next tile when kt+2 >= nk) is issued after the extras; the wait then lets exactly those younger operations stay
        // in flight, so slice kt+1 -- and everything older, the fetches of earlier iterations included -- has landed.
        // FX (counter scheduling only): 1 = wave 0 draws the next position, 2 = it publishes the answer in LDS, 3 = everyone reads it
        auto kstep = [&](int kt, auto ST, auto LD, auto FX) __attribute__((always_inline)) {
            constexpr int st_g = decltype(ST)::value, ld_g = decltype(LD)::value, fx = decltype(FX)::value;
            constexpr int s_lo = st_g < 0 ? 0 : st_g * NSTORE / HEAD7, s_hi = st_g < 0 ? 0 : (st_g + 1) * NSTORE / HEAD7;
            constexpr int l_lo = ld_g < 0 ? 0 : ld_g * NLOAD / TAIL7, l_hi = ld_g < 0 ? 0 : (ld_g == TAIL7 - 1 ? NLOAD + NCONST : (ld_g + 1) * NLOAD / TAIL7);
            const uint32_t stg = lds_base + ring * STAGE7;
            uint4 wf0[4], af0[4], wf1[4], af1[4];
            V7_READ4(wf0, stg + w_off0);
            V7_READ4(af0, stg + a_off0);
            V7_READ4(wf1, stg + (w_off0 ^ 64u));
            V7_READ4(af1, stg + (a_off0 ^ 64u));
            if (fx == 3 && dyn) asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(slot_addr) : "memory");  // waited for with the fragments below
            // (`pending` already says that the PREVIOUS tile's columns were inside N for this wave; this tile's `live` has nothing to
            // do with it -- consecutive tiles of a workgroup sit in different column tiles since round 4's drawn order)
            const bool do_st = s_hi > s_lo && pending;
            if (do_st) {
#pragma unroll
                for (int sl = s_lo; sl < s_hi; ++sl) store_slot(sl, false);
            }
            if (l_hi > l_lo) {
#pragma unroll
                for (int sl = l_lo; sl < l_hi; ++sl) load_slot(sl);
            }
            const bool draw = fx == 1 && dyn && wave == 0;  // wave-uniform (scalar) condition: one more operation on wave 0's counter
            if (draw) sched_draw(fetched, ctr);
            asm volatile("" ::: "memory");  // extras stay in front of this iteration's LDS-DMA pieces
            const int stage2 = ring == 0 ? 2 : ring - 1;  // (ring + 2) % 3
            bool issued = true;
            if (kt + 2 < nk) issue(false, kt + 2, stage2);
            else if (has_next) issue(true, kt + 2 - nk, stage2);
            else issued = false;
#define V7_WAIT(n)                                                                  \
    do {                                                                            \
        if (draw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) + 1) : "memory");    \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");               \
    } while (0)
            if (issued) {
                if (s_hi > s_lo) {
                    if (do_st) V7_WAIT(PIECES7 + (s_hi - s_lo));
                    else V7_WAIT(PIECES7);
                } else {
                    V7_WAIT(PIECES7 + (l_hi - l_lo));
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // last tile of this workgroup, last two slices: nothing to keep in flight
            }
#undef V7_WAIT
            if (fx == 2 && dyn && wave == 0) {  // iteration 1's wait retired the draw of iteration 0: hand it to the other waves
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
                for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc[ni][mi], wf0[ni], af0[mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc[ni][mi], wf1[ni], af1[mi]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            ring = ring == 2 ? 0 : ring + 1;
        };

        static_assert(HEAD7 >= 3, "the counter's answer travels through iterations 0-2");
        kstep(0, IC<0>(), IC<-1>(), IC<1>());
        kstep(1, IC<1>(), IC<-1>(), IC<2>());
        kstep(2, IC<2>(), IC<-1>(), IC<3>());
        if (dyn) {
            next = __builtin_amdgcn_readfirstlane((int)seen);  // the counter's value before this workgroup's add = the position drawn
            has_next = next < xcnt;
            reset_ctr = reset_ctr || next == xcnt + xgrid - 1;  // nobody draws after the last of the xcnt + xgrid draws
        }
        if (has_next) {  // where the next tile's operands are; needed from iteration nk-2 on
            pos_origin(next, nm0, nn0);
#pragma unroll
            for (int j = 0; j < PIECES7; ++j) srcn[j] = piece_offset(j, nm0, nn0);
        }
#define V7_HEAD(i) \
    if constexpr (HEAD7 > i) kstep(i, IC<i>(), IC<-1>(), IC<0>());
        V7_HEAD(3) V7_HEAD(4) V7_HEAD(5) V7_HEAD(6) V7_HEAD(7)
#undef V7_HEAD
        for (int kt = HEAD7; kt < nk - TAIL7 - 1; ++kt) kstep(kt, IC<-1>(), IC<-1>(), IC<0>());
        // a fetch group only exists if it has something to fetch (the last one also carries the constants)
#define V7_TAIL(i) \
    if constexpr (TAIL7 > i) kstep(nk - 1 - TAIL7 + i, IC<-1>(), IC<((i == TAIL7 - 1 ? NLOAD + NCONST : NLOAD) > 0) ? i : -1>(), IC<0>());
        V7_TAIL(0) V7_TAIL(1) V7_TAIL(2) V7_TAIL(3) V7_TAIL(4) V7_TAIL(5) V7_TAIL(6) V7_TAIL(7)
#undef V7_TAIL
        kstep(nk - 1, IC<-1>(), IC<-1>(), IC<0>());  // its wait retired the fetches of the iterations before it

        // ---- epilogue arithmetic (gemm_epilogue_fast's, result kept in registers); one barrier interval of its own: the other
        // wave group runs its MFMA phase beside it ----
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float v[16];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[ni * 4 + r] = (F & F_BIAS) ? acc[ni][mi][r] + __uint_as_float(reinterpret_cast<const uint32_t*>(&bias4[ni])[r]) : acc[ni][mi][r];
            if ((F & F_GELU) && !TWO) Gelu<T>::fwd16(v);  // TWO: the pre-activation is kept, GELU is applied when it is stored
            if (F & F_GELU_BWD) {
                float x[16];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    Vec16<T> t;
                    t.raw = pa[mi][h];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[h * 8 + j] = t.get(j);
                }
                if (p.act == LNX_ACT_MUL_AUX) {  // the forward saved GELU'(pre-activation): one multiply
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] *= x[j];
                } else {
                    Gelu<T>::mulgrad16(v, x);
                }
            }
            if (F & F_RES) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sc = p.rowscale != nullptr ? rs[mi] : 1.0f;  // (without a row scale the fetch read a bias entry)
                    v[4 * q + 0] = fmaf(v[4 * q + 0], sc, __uint_as_float(pr[mi][q].x));
                    v[4 * q + 1] = fmaf(v[4 * q + 1], sc, __uint_as_float(pr[mi][q].y));
                    v[4 * q + 2] = fmaf(v[4 * q + 2], sc, __uint_as_float(pr[mi][q].z));
                    v[4 * q + 3] = fmaf(v[4 * q + 3], sc, __uint_as_float(pr[mi][q].w));
                }
            }
            if (OUT_F32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) hf[mi][q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else {
                Vec16<T> o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o.set(j, v[h * 8 + j]);
                    hc[mi][h] = o.raw;
                }
            }
        }
        pend_row = live ? mbase : -1;
        pend_col = nb;
        if (pend_row >= 0 && (!full || !has_next)) {
            // a partial tile (rows beyond M: some lanes store nothing, the counted waits need every lane to) and the last
            // tile (no K loop left to hide behind) store at once; the next iteration's wait then also covers these stores
#pragma unroll
            for (int sl = 0; sl < NSTORE; ++sl) store_slot(sl, true);
            pend_row = -1;
        }
        if (!has_next) break;  // (waves 4-7 leave one barrier short: they made one more at the start)
        __builtin_amdgcn_s_barrier();
        pos = next;
        m0 = nm0;
        n0 = nn0;
        src = srcn;
    }
    if (reset_ctr && tid == 0) sched_reset(ctr);
}

bool nt_v7_ok(const GemmP& p, int f, bool out_f32) {
    if (f == (int)F_GENERIC || p.a_mode == LNX_ADDR_PATCH2) return false;
    if (p.act == LNX_ACT_GELU_D) return false;  // this kernel's fc1 form keeps ONE tensor per tile; the derivative form needs two
    if (p.K % BK7 != 0 || p.K / BK7 < 6) return false;
    if (p.N % 64 != 0) return false;  // a wave's 64 columns are all inside or all outside N
    const int64_t lim = (int64_t)1 << 31;  // 32-bit byte offsets of the LDS-DMA sources
    if ((int64_t)p.M * p.lda * 2 >= lim || (int64_t)p.N * p.ldw * 2 >= lim) return false;
    if (out_f32) return f == (F_BIAS | F_RES);
    return f == 0 || f == F_BIAS || f == (F_BIAS | F_C2 | F_GELU) || f == F_GELU_BWD;
}

int launch_nt_v7(const GemmP& p0, int f, bool out_f32, hipStream_t st) {
    GemmP p = p0;
    p.tiles_m = cdiv(p.M, BM7);
    p.tiles_n = cdiv(p.N, BN7);
    const int ntiles = p.tiles_m * p.tiles_n;
    const int cus = device_cus();
    if (cus <= 0) return 1;
    const int room = persistent_cus(cus);              // cus - LNX_CU_MARGIN / lnx_set_cu_margin()
    const int grid = ntiles < room ? ntiles : room;    // persistent: one workgroup per CU (144 KiB of LDS each)
    const size_t lds = NST7 * STAGE7 + 16;             // + the dword the next tile's position is published through
    p.tile_slot = tile_sched_static() ? -1 : tile_slot_of(st);
    const bool deep = p.K / BK7 >= 17;  // long K loops spread the deferred traffic over 8 + 8 iterations instead of 3 + 2
#define V7_LAUNCH_(O, FF, H, T)                                                                                                      \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v7_kernel<O, FF, H, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_v7_kernel<O, FF, H, T>), dim3(grid), dim3(512), lds, st, p);                                     \
    } while (0)
#define V7_LAUNCH(O, FF)                   \
    do {                                   \
        if (deep) V7_LAUNCH_(O, FF, 8, 8); \
        else V7_LAUNCH_(O, FF, 3, 2);      \
    } while (0)
    if (out_f32) V7_LAUNCH(true, F_BIAS | F_RES);
    else if (f == 0) V7_LAUNCH(false, 0);
    else if (f == F_BIAS) V7_LAUNCH(false, F_BIAS);
    else if (f == (F_BIAS | F_C2 | F_GELU)) V7_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
    else V7_LAUNCH(false, F_GELU_BWD);
#undef V7_LAUNCH
#undef V7_LAUNCH_
    return 0;
}

}  // namespace lnxg
