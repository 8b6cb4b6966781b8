// Stand-in for a ring all-reduce on ONE GPU: a kernel of `nwg` workgroups that streams a gradient bucket
// HBM -> HBM for a given wall-clock duration, launched on the communication stream where linnaeus_amd/ddp.py issues the bucket's
// all-reduce.  What it reproduces of an RCCL collective kernel: a few dozen long-lived workgroups that own their CUs' LDS (so a
// 144-KiB-LDS GEMM workgroup cannot be placed beside them) and a steady stream of HBM reads + writes.  What it does not: xGMI.
// Every workgroup leaves when the 100 MHz wall clock says so: the grid always drains.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/ubench/cu_hog.hip -o tools/libcu_hog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void hog_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, unsigned long long ticks) {
    extern __shared__ unsigned char lds[];  // only claimed, to keep LDS-hungry workgroups off this CU (as an RCCL kernel's buffers do)
    const unsigned long long t0 = wall_clock64();
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    if (threadIdx.x == 0) lds[0] = 1;
    for (int pass = 0; pass < (1 << 20); ++pass) {  // bounded twice over: by the clock and by a pass count
        for (size_t i = lo + threadIdx.x; i < hi; i += 4 * 256) {
            uint4 a = src[i], b = i + 256 < hi ? src[i + 256] : a, c = i + 512 < hi ? src[i + 512] : a, d = i + 768 < hi ? src[i + 768] : a;
            a.x += b.x; c.x += d.x;  // "reduce"
            dst[i] = a;
            if (i + 256 < hi) dst[i + 256] = b;
            if (i + 512 < hi) dst[i + 512] = c;
            if (i + 768 < hi) dst[i + 768] = d;
            if (wall_clock64() - t0 >= ticks) return;
        }
        if (wall_clock64() - t0 >= ticks) return;
    }
}

// streams `bytes` from src to dst (16-byte aligned, may alias) over and over with `nwg` workgroups for `us` microseconds
extern "C" int hog_copy(const void* src, void* dst, size_t bytes, int nwg, double us, int lds_bytes, void* stream) {
    if (nwg <= 0 || bytes < 16 || us <= 0) return 1;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(hog_kernel, dim3(nwg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, bytes / 16, (unsigned long long)(us * 100.0));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
