// Operand / scale / result layout of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) on gfx950, checked with exact data against
// the hypothesis (the 16x16x128 instruction's pattern, tools/ubench/mfma_mx_map.hip, carried over):  lane l = (r = l & 31,
// h = l >> 5) holds row r of A (column r of B); byte j of its 32 operand bytes pairs with byte j of the B lane with the same h
// (any k order does, as long as A and B agree); MX block kb of row r = dwords 4 kb .. 4 kb + 3 of BOTH lanes r and r + 32, and
// its E8M0 scale is byte 0 of lane (r + 32 kb)'s scale register; result register i of lane l is
// C[row = 8 (i / 4) + 4 (l >> 5) + i % 4][col = l & 31].
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mx32_layout.hip -o /tmp/mx32 && /tmp/mx32
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(16))) float v16f;

__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* out) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = reinterpret_cast<const int*>(A + l * 32)[i];
        b[i] = reinterpret_cast<const int*>(B + l * 32)[i];
    }
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i];
}

static float e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    static const unsigned char vals[5] = {0x00, 0x38, 0x40, 0x30, 0x3C};  // 0, 1, 2, 0.5, 1.5
    unsigned char hA[64 * 32], hB[64 * 32], hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, kh = l >> 5;
        hsa[l] = (unsigned char)(127 + (l % 7) - 3);   // 2^-3 .. 2^3, a different one per lane
        hsb[l] = (unsigned char)(127 + ((l * 5) % 4) - 1);
        for (int j = 0; j < 32; ++j) {
            const int k = 32 * kh + j;
            hA[l * 32 + j] = vals[(r + 2 * k) % 5];
            hB[l * 32 + j] = vals[(3 * r + k) % 5];
        }
    }
    unsigned char *dA, *dB, *dsa, *dsb;
    float* dout;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dsa, 64); (void)hipMalloc(&dsb, 64); (void)hipMalloc(&dout, 64 * 16 * 4);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dout);
    static float h[64 * 16];
    (void)hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    double maxd = 0;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i / 4) + 4 * (l >> 5) + i % 4, col = l & 31;
            double ref = 0;
            for (int hh = 0; hh < 2; ++hh)
                for (int j = 0; j < 32; ++j) {
                    const int kb = j / 16;
                    ref += (double)e4m3(hA[(row + 32 * hh) * 32 + j]) * ldexp(1.0, hsa[row + 32 * kb] - 127) * e4m3(hB[(col + 32 * hh) * 32 + j]) * ldexp(1.0, hsb[col + 32 * kb] - 127);
                }
            const double d = fabs(ref - h[l * 16 + i]);
            if (d > 1e-3 && bad++ < 8) printf("lane %d reg %d (row %d col %d): got %g expected %g\n", l, i, row, col, h[l * 16 + i], ref);
            if (d > maxd) maxd = d;
        }
    printf("32x32x64 f8f6f4 layout hypothesis: max |difference| = %g over 1024 outputs -> %s\n", maxd, maxd < 1e-3 ? "CONFIRMED" : "WRONG");
    return 0;
}
