// How many bytes per clock can one CU pull from its XCD's L2 -- into LDS by LDS-DMA, into registers by global_load_dwordx4,
// and by both at once?  (The GEMM K loops are bound by this feed: a 256x128 bf16 tile needs 47 B/clk/CU at full MFMA rate, a
// 256x256 tile 32.)  One 512-thread workgroup per CU, every workgroup sweeps its own 96 KiB window of a buffer over and over
// (L2-resident after the first sweep; windows of the workgroups of an XCD together stay below 4 MiB).
// Build: hipcc -O3 --offload-arch=gfx950 l2_feed.hip -o l2_feed
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int WIN = 96 * 1024;  // bytes per workgroup window

template <int MODE>  // 0 = LDS-DMA only, 1 = register loads only, 2 = half of the bytes each way, 3 = both at full rate (2x bytes)
__global__ __launch_bounds__(512) void k(const unsigned char* buf, int iters, unsigned* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* win = buf + (size_t)blockIdx.x * WIN;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // one sweep = 96 pieces of 1 KiB: wave w takes pieces w, w + 8, ...
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int piece = wave + 8 * j;
            const unsigned char* src = win + piece * 1024 + lane * 16;
            const bool dma = MODE == 0 || MODE == 3 || (MODE == 2 && (j & 1) == 0);
            const bool reg = MODE == 1 || MODE == 3 || (MODE == 2 && (j & 1) == 1);
            if (dma)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (piece % 48) * 1024), 16, 0, 0);
            if (reg) {
                uint4 v;
                const unsigned char* rsrc = MODE == 3 ? win + ((piece + 48) % 96) * 1024 + lane * 16 : src;  // the other half of the window
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(rsrc) : "memory");
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // keep a dozen in flight; data of older ones has landed
                acc += v.x;
            }
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
static void run(const unsigned char* buf, unsigned* sink, unsigned long long* cyc, int nwg) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(512), 48 * 1024, 0, buf, 20, sink, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(nwg), dim3(512), 48 * 1024, 0, buf, iters, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < nwg; ++i) mean += (double)h[i];
    mean /= nwg;
    const double bytes = (double)WIN * iters * (MODE == 3 ? 2 : 1);
    const char* names[] = {"LDS-DMA only", "register loads only", "half / half", "both, 2x bytes"};
    printf("%-20s %6.1f B/clk/CU (in-kernel cycles)   %6.2f TB/s chip (%d workgroups, %.0f us)\n", names[MODE], bytes / mean, bytes * nwg / (ms * 1e-3) / 1e12, nwg,
           ms * 1e3);
}

int main() {
    const int nwg = 256;
    unsigned char* buf;
    unsigned* sink;
    unsigned long long* cyc;
    hipMalloc(&buf, (size_t)nwg * WIN);
    hipMalloc(&sink, 4);
    hipMalloc(&cyc, 256 * 8);
    hipMemset(buf, 1, (size_t)nwg * WIN);
    run<0>(buf, sink, cyc, nwg);
    run<1>(buf, sink, cyc, nwg);
    run<2>(buf, sink, cyc, nwg);
    run<3>(buf, sink, cyc, nwg);
    return 0;
}
