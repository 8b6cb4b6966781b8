// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales (E8M0 127).
// Hypothesis under test: lane l holds A[row l & 15][k = 32 (l >> 4) + j] in byte j of its 8 dwords, B likewise with the
// column on l & 15, C/D as every other 16x16 MFMA.  Exact small-integer data; prints the number of wrong elements.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f8_layout.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C) {  // A [16][128], B [16][128] (n-major), C [16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    for (int d = 0; d < 8; ++d) {
        a[d] = *reinterpret_cast<const int*>(A + r * 128 + 32 * g + 4 * d);
        b[d] = *reinterpret_cast<const int*>(B + r * 128 + 32 * g + 4 * d);
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const int one = 0x7f7f7f7f;  // E8M0 127 = 2^0 in every byte
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, one, 0, one);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];  // row 4g+i, col r
}

static uint8_t enc(int v) {  // e4m3fn of a small integer in [-4, 4]
    static const uint8_t tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};  // 0, 1, 2, 3, 4
    return v >= 0 ? tab[v] : (uint8_t)(0x80 | tab[-v]);
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<int> a(16 * 128), b(16 * 128);
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) {
        a[i] = rand() % 9 - 4; b[i] = rand() % 9 - 4;
        A[i] = enc(a[i]); B[i] = enc(b[i]);
    }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    float C[256];
    hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            int ref = 0;
            for (int k = 0; k < 128; ++k) ref += a[m * 128 + k] * b[n * 128 + k];
            if ((float)ref != C[m * 16 + n]) { if (bad < 4) printf("C[%d][%d] = %g, expected %d\n", m, n, C[m * 16 + n], ref); ++bad; }
        }
    printf("wrong elements: %d of 256\n", bad);
    return 0;
}
