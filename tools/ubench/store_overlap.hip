// Can a CU keep pulling operands from L2 (LDS-DMA, as a GEMM K loop does) while OTHER waves of the same workgroup stream an
// epilogue's stores to HBM?  Stores, loads and LDS-DMA of ONE wave share an in-order counter (a K loop that stores waits for
// its stores: DESIGN.md 8a); this asks whether moving the stores to different waves frees the loads, or whether the CU's
// shared vector-memory path (address unit -> L1 -> L2 queues) couples them anyway once HBM writes back up.
// One 512-thread workgroup per CU: waves 0-3 sweep a 96-KiB L2-resident window by LDS-DMA (MODE 0, 2), waves 4-7 write
// fresh 16-byte-per-lane rows of a large buffer (MODE 1, 2).  Reports the loaders' B/clk/CU and the chip's store rate.
// Build: hipcc -O3 --offload-arch=gfx950 store_overlap.hip -o store_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int WIN = 96 * 1024;

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* win_buf, unsigned char* out, size_t out_per_wg, int iters, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (MODE != 1) {
            const unsigned char* win = win_buf + (size_t)blockIdx.x * WIN;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 24; ++j) {  // 96 pieces of 1 KiB over 4 waves
                    const int piece = wave + 4 * j;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(win + piece * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(smem + (piece % 48) * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    } else {
        if (MODE != 0) {
            unsigned char* o = out + (size_t)blockIdx.x * out_per_wg + (size_t)(wave - 4) * 1024 + lane * 16;
            const uint4 v = make_uint4(lane, wave, blockIdx.x, 7u);
            // the same bytes per iteration as the loaders move: 96 KiB per workgroup and iteration
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 24; ++j) *reinterpret_cast<uint4*>(o + (size_t)(it * 24 + j) * 4096) = v;
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 256) cycles[256 + blockIdx.x] = t1 - t0;
    }
}

template <int MODE>
static void run(const unsigned char* win, unsigned char* out, size_t out_per_wg, unsigned long long* cyc, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 48 * 1024, 0, win, out, out_per_wg, 8, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 48 * 1024, 0, win, out, out_per_wg, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double ml = 0, msr = 0;
    for (int i = 0; i < 256; ++i) { ml += (double)h[i]; msr += (double)h[256 + i]; }
    ml /= 256; msr /= 256;
    const double bytes = (double)WIN * iters;
    const char* names[] = {"loads only", "stores only", "loads + stores"};
    printf("%-15s kernel %7.1f us", names[MODE], ms * 1e3);
    if (MODE != 1) printf("   loaders %5.1f B/clk/CU (%6.0f cycles)", bytes / ml, ml);
    if (MODE != 0) printf("   stores %5.2f TB/s over the kernel, store waves done after %6.0f cycles", bytes * 256 / (ms * 1e-3) / 1e12, msr);
    printf("\n");
}

int main() {
    const int iters = 256;  // 24 MiB of stores per workgroup, 6 GiB over the chip
    const size_t out_per_wg = (size_t)iters * 24 * 4096;
    unsigned char *win, *out;
    unsigned long long* cyc;
    hipMalloc(&win, (size_t)256 * WIN);
    hipMalloc(&out, out_per_wg * 256);
    hipMalloc(&cyc, 512 * 8);
    hipMemset(win, 1, (size_t)256 * WIN);
    hipMemset(out, 0, out_per_wg * 256);
    run<0>(win, out, out_per_wg, cyc, iters);
    run<1>(win, out, out_per_wg, cyc, iters);
    run<2>(win, out, out_per_wg, cyc, iters);
    run<0>(win, out, out_per_wg, cyc, iters);
    return 0;
}
