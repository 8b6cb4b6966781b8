// Scale-operand probe for v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, E8M0 block scales).
// Established with tools/ubench/mfma_mx_map.hip: a lane's dwords 0-3 are k = 16 g .. 16 g + 15 and its dwords 4-7 are
// k = 64 + 16 g .. 64 + 16 g + 15 (g = l >> 4), NOT 32 contiguous elements; the MX blocks are the natural 32-element
// blocks of that k.  Hypothesis under test here: the scale of block kb of row r comes from lane 16 kb + r, from the byte of
// its scale register that op_sel (0..3) selects; the other three bytes are ignored.  Exact small-integer data and power-of-two scales; prints the number
// of wrong elements for op_sel 0 and 2.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mx_scale.hip -o /tmp/mxprobe && /tmp/mxprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

template <int SEL>
__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* C) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    for (int d = 0; d < 8; ++d) {
        const int k = 64 * (d >> 2) + 16 * g + 4 * (d & 3);
        a[d] = *reinterpret_cast<const int*>(A + r * 128 + k);
        b[d] = *reinterpret_cast<const int*>(B + r * 128 + k);
    }
    // own scale in byte SEL, junk (E8M0 140 / 90) in the others
    const int sa = (0x8c5a8c5a & ~(0xff << (8 * SEL))) | ((int)SA[r * 4 + g] << (8 * SEL));
    const int sb = (0x5a8c5a8c & ~(0xff << (8 * SEL))) | ((int)SB[r * 4 + g] << (8 * SEL));
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, SEL, sa, SEL, sb);
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}

static uint8_t enc(int v) {
    static const uint8_t tab[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return v >= 0 ? tab[v] : (uint8_t)(0x80 | tab[-v]);
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128), SA(64), SB(64);
    std::vector<int> a(16 * 128), b(16 * 128);
    srand(5);
    for (int i = 0; i < 16 * 128; ++i) {
        a[i] = rand() % 9 - 4; b[i] = rand() % 9 - 4;
        A[i] = enc(a[i]); B[i] = enc(b[i]);
    }
    for (int i = 0; i < 64; ++i) { SA[i] = 127 + rand() % 7 - 3; SB[i] = 127 + rand() % 7 - 3; }
    uint8_t *dA, *dB, *dSA, *dSB; float* dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
    for (int sel = 0; sel < 4; sel += 2) {
        if (sel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
        else hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
        float C[256];
        hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int kb = 0; kb < 4; ++kb) {
                    int s = 0;
                    for (int k = 0; k < 32; ++k) s += a[m * 128 + 32 * kb + k] * b[n * 128 + 32 * kb + k];
                    ref += (double)s * exp2((double)(SA[m * 4 + kb] - 127) + (double)(SB[n * 4 + kb] - 127));
                }
                if ((float)ref != C[m * 16 + n]) { if (bad < 4) printf("sel %d: C[%d][%d] = %g, expected %g\n", sel, m, n, C[m * 16 + n], ref); ++bad; }
            }
        printf("op_sel %d: wrong elements: %d of 256\n", sel, bad);
    }
    return 0;
}
