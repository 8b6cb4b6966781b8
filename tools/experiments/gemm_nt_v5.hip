// gemm_nt v5 -- MEASUREMENT KERNEL, not part of the product library (moved out of linnaeus_amd/csrc/gemm2.hip in round 4).
// It is the vehicle of the "convoy" study: same time as the 8-wave kernel on every form.  Build with
// tools/experiments/build.sh and load with LNX_LIB_PATH=tools/liblnx_experiments.so; LNX_NT_V5=1 then routes every NT product
// it can run through it (LNX_V5_STAGGER=n delays every other workgroup by n K-slices; -DV5_ABLATE adds the phase stamps).
#include "gemm_common.hpp"

namespace lnxg {

#define DS4_READ4(dst, addr)                                                                            \
    do {                                                                                                \
        const uint32_t a_ = (addr);                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(a_) : "memory");                        \
        asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(dst[1]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(dst[2]) : "v"(a_) : "memory");             \
        asm volatile("ds_read_b128 %0, %1 offset:768" : "=v"(dst[3]) : "v"(a_) : "memory");             \
    } while (0)
constexpr int BK4 = 32, ROWB4 = 64;  // 32-element K slices in 64-byte LDS rows, as gemm_nt_v4
__device__ __forceinline__ int key4r(int row) { return (row & 16) ? 3 : 0; }

// ------------------------------------------------------------------------------------
// gemm_nt v5: the in-model kernel for products whose epilogue moves as many bytes as the K loop (round 3).
// The 8-wave kernels above hold the whole CU (144 / 128 KiB of LDS, one workgroup), so a tile's life is prologue
// (first slices: exposed L2/HBM latency) -> K loop -> epilogue (loads of aux / residual, 1-2 stores per element), one
// after the other: at K = 384 the K loop is a third of it, and fc1's epilogue alone (two [M, hid] stores) is HBM time
// nothing runs beside.  Here a workgroup is FOUR waves on a 256x128 tile (each wave 128x64 = the v4 wave tile, 128
// accumulator registers), K in 32-element slices through a 3-deep ring of 24 KiB stages = 72 KiB: TWO workgroups share a
// CU (2 waves per SIMD, 256 registers each), and the hardware interleaves them -- one workgroup's epilogue stores and
// prologue fetches run under the other's MFMAs.  No ping-pong wave groups: the partner on each SIMD is the other
// workgroup's wave.  One barrier per slice:
//   iteration t: fragment reads of stage t%3 ; issue slice t+2 -> stage (t+2)%3 ; wait (own pieces of t+1 landed, reads
//   done) ; barrier ; 32 MFMAs.
// WAR: stage (t+2)%3 was last read in iteration t-1 and every wave retired those reads (lgkmcnt(0)) before barrier t-1.
// ------------------------------------------------------------------------------------
// Round 5: BM5 = 128 as a second instantiation (LNX_V5_BM=128): a wave is then one 64x64 tile, a stage 16 KiB, and a product with N = 384
// has 597 tiles at 128 images for the chip's 512 workgroup slots instead of 300 tiles of 256x128 for 256 CUs.
constexpr int BN5 = 128;
constexpr int NST5 = 3;

template <bool OUT_F32, int F, int BM5>
__global__ __launch_bounds__(256, 2) void gemm_nt_v5_kernel(const GemmP p) {
    constexpr int STAGE5 = (BM5 + BN5) * ROWB4;   // 24 KiB (16 KiB at BM5 = 128)
    constexpr int PIECES5 = STAGE5 / 1024 / 4;    // 1-KiB LDS-DMA instructions per wave per stage = 6 (4)
    constexpr int WROWS = BM5 / 2;                // rows of a wave tile: 128 (two 64-row accumulator sets) or 64
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST5][A 256 rows | W 128 rows][64 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int s = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tn = logical % p.tiles_n;
    const int tm = logical / p.tiles_n;
    const int m0 = tm * BM5, n0 = tn * BN5;

    // piece i (24 per stage) fills LDS rows 16 i .. 16 i + 15 (rows 0..255 = A, 256..383 = W); wave w issues i = w + 4 j
    const unsigned char* src[PIECES5];
#pragma unroll
    for (int j = 0; j < PIECES5; ++j) {
        const int i = wave + 4 * j;
        const int row = 16 * i + (lane >> 2);
        const int slot = lane & 3;
        if (row < BM5) {
            const int chunk = slot ^ key4r(row);
            int m = m0 + row;
            if (m >= p.M) m = p.M - 1;
            src[j] = p.A + ((int64_t)m * p.lda + chunk * 8) * 2;
        } else {
            const int wr = row - BM5;
            const int chunk = slot ^ key4r(wr);
            int n = n0 + wr;
            if (n >= p.N) n = p.N - 1;
            src[j] = p.W + ((int64_t)n * p.ldw + chunk * 8) * 2;
        }
    }
    auto issue_stage = [&](int kt, int stage) {
#pragma unroll
        for (int j = 0; j < PIECES5; ++j) {
            const int i = wave + 4 * j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (int64_t)kt * BK4 * 2),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE5 + i * 1024), 16, 0, 0);
        }
    };

    const int frag_row = (s >> 2) * 16 + (s & 3);
    const uint32_t chunk_off = (uint32_t)((g ^ (((s >> 2) & 1) * 3)) << 4);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t a_off = (uint32_t)((wm * WROWS + frag_row) * ROWB4) + chunk_off;
    const uint32_t w_off = (uint32_t)((BM5 + wn * 64 + frag_row) * ROWB4) + chunk_off;

    f32x4_t acc0[4][4], acc1[4][4];  // rows 0..63 / 64..127 of the wave tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }

    int nk = p.K / BK4;  // >= 2
#ifdef V5_ABLATE
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(p.C8s);  // diagnostic: per-workgroup phase times (100 MHz clock)
    if (stamp && tid == 0) {
        stamp[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        stamp[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
#ifdef V5_ABLATE  // stagger bit 17 = two K slices only (the epilogue alone, more or less)
    if (p.stagger & (1 << 17)) nk = 2;
#endif
    if ((p.stagger & 0xffff) > 0) {
        // the two workgroups of a CU start together and, being identical, stay in step (both in the K loop, then both in
        // the epilogue: nothing overlaps).  The one in the odd wave slot of its SIMD starts late once; workgroups that
        // later take over a freed slot inherit the offset.
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);  // HW_REG_HW_ID[3:0] = wave slot within the SIMD
        if (slot & 1)
            for (int i = 0; i < (p.stagger & 0xffff); ++i) __builtin_amdgcn_s_sleep(16);  // ~1024 cycles per step
    }
    issue_stage(0, 0);
    issue_stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES5) : "memory");  // own pieces of slice 0 have landed
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t st = lds_base + (kt % NST5) * STAGE5;
        uint4 wf[4], af0[4], af1[4];
        DS4_READ4(wf, st + w_off);
        DS4_READ4(af0, st + a_off);
        if constexpr (BM5 == 256) DS4_READ4(af1, st + a_off + 64 * ROWB4);
        if (kt + 2 < nk) {
            issue_stage(kt + 2, (kt + 2) % NST5);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES5) : "memory");
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
        if constexpr (BM5 == 256) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mfma<T>::run(acc1[ni][mi], wf[ni], af1[mi]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef V5_ABLATE
    if (stamp && tid == 0) stamp[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef V5_ABLATE  // diagnostic build (tools/build_v5_ablate.sh): stagger bit 16 = no epilogue (stores only if an accumulator holds a magic value)
    if ((p.stagger & (1 << 16)) && acc0[0][0][0] != 1.2345e30f && acc1[3][3][3] != 1.2345e30f) return;
#endif
    gemm_epilogue_fast<T, OUT_F32, F>(p, acc0, m0 + wm * WROWS, n0 + wn * 64, lane);
    if constexpr (BM5 == 256) gemm_epilogue_fast<T, OUT_F32, F>(p, acc1, m0 + wm * 128 + 64, n0 + wn * 64, lane);
#ifdef V5_ABLATE
    if (stamp && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


bool nt_v5_ok(const GemmP& p, int f) {
    if (f == (int)F_GENERIC || p.a_mode == LNX_ADDR_PATCH2) return false;
    return p.K % BK4 == 0 && p.K >= 2 * BK4;
}

template <int BM5>
static int launch_nt_v5_t(const GemmP& p0, int f, bool out_f32, hipStream_t st) {
    GemmP p = p0;
    p.tiles_m = cdiv(p.M, BM5);
    p.tiles_n = cdiv(p.N, BN5);
    const char* sg = getenv("LNX_V5_STAGGER");
    p.stagger = sg ? atoi(sg) : 0;
#ifdef V5_ABLATE
    const char* sp = getenv("LNX_V5_STAMPS");  // device address of a [grid][4] uint64 buffer
    p.C8s = sp ? reinterpret_cast<unsigned char*>(strtoull(sp, nullptr, 10)) : nullptr;
#endif
    const int grid5 = p.tiles_m * p.tiles_n;
    size_t lds5 = NST5 * (BM5 + BN5) * ROWB4;
#ifdef V5_ABLATE
    if (getenv("LNX_V5_ONE_WG")) lds5 = 100 * 1024;  // diagnostic: one workgroup per CU
#endif
#define V5_LAUNCH(O, FF)                                                                                                             \
    do {                                                                                                                             \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_v5_kernel<O, FF, BM5>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_v5_kernel<O, FF, BM5>), dim3(grid5), dim3(256), lds5, st, p);                                    \
    } while (0)
    if (out_f32) V5_LAUNCH(true, F_BIAS | F_RES);
    else if (f == 0) V5_LAUNCH(false, 0);
    else if (f == F_BIAS) V5_LAUNCH(false, F_BIAS);
    else if (f == (F_BIAS | F_C2 | F_GELU)) V5_LAUNCH(false, F_BIAS | F_C2 | F_GELU);
    else if (f == (F_BIAS | F_GELU)) V5_LAUNCH(false, F_BIAS | F_GELU);
    else V5_LAUNCH(false, F_GELU_BWD);
#undef V5_LAUNCH
    return 0;
}

int launch_nt_v5(const GemmP& p0, int f, bool out_f32, hipStream_t st) {
    const char* bm = getenv("LNX_V5_BM");  // 128: the half-height tile (round 5)
    if (bm && atoi(bm) == 128) return launch_nt_v5_t<128>(p0, f, out_f32, st);
    return launch_nt_v5_t<256>(p0, f, out_f32, st);
}

}  // namespace lnxg
