// What does HBM deliver to plain streaming kernels on this chip -- reads, writes, and a copy -- on buffers far larger than the
// 256 MB Infinity Cache?  (The step's own roofline: DESIGN.md section 6.)  16 bytes per lane, 8 accesses in flight per lane,
// grid-stride over 4 GiB.
// Build: hipcc -O3 --offload-arch=gfx950 hbm_rw.hip -o hbm_rw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>  // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = MODE == 1 ? make_uint4((unsigned)i, j, 3u, 4u) : in[i + j * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) acc += v[j].x ^ v[j].w;
            else out[i + j * stride] = v[j];
        }
    }
    if (MODE == 0 && acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
static void run(const uint4* in, uint4* out, size_t n16, unsigned* sink, int wgs) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, in, out, n16, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, in, out, n16, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double bytes = (double)n16 * 16 * (MODE == 2 ? 2 : 1);
    const char* names[] = {"read ", "write", "copy "};
    printf("%s %4d workgroups: %7.1f us  %5.2f TB/s%s\n", names[MODE], wgs, ms * 1e3, bytes / (ms * 1e-3) / 1e12, MODE == 2 ? " (read + written bytes)" : "");
}

int main() {
    const size_t bytes = (size_t)4 << 30, n16 = bytes / 16;
    uint4 *a, *b;
    unsigned* sink;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    for (int wgs : {256, 512, 1024, 2048, 8192}) {
        run<0>(a, b, n16, sink, wgs);
        run<1>(a, b, n16, sink, wgs);
        run<2>(a, b, n16, sink, wgs);
    }
    return 0;
}
