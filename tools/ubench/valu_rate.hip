// VALU issue-rate microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32, W waves per SIMD, independent chains.
// Prints SIMD cycles per wave-instruction (shader clock assumed 2.4 GHz) -- settles whether packed fp32 doubles throughput.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int PK>
__global__ void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, a8 = 8, a9 = 9, a10 = 10, a11 = 11, a12 = 12, a13 = 13, a14 = 14, a15 = 15;
    const float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}, p4{a8, a9}, p5{a10, a11}, p6{a12, a13}, p7{a14, a15};
    const f2 bb{b, b}, cc{c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (PK) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(bb), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(bb), "v"(cc));
            } else {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int w = 1; w <= 8; w *= 2) {
            // one 256-thread workgroup = one wave per SIMD; w workgroups per CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(256 * w), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<0>, dim3(256 * w), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * w;
            printf("%s  %d wave(s)/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", w, ms,
                   ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    return 0;
}
