// Which lane's scale byte governs which operand bytes of v_mfma_scale_f32_16x16x128_f8f6f4?
// Experiment (g, d, gs): A is 1.0 in dword d of the lanes of group g (l >> 4 == g) and 0 elsewhere, B is all ones, the
// A-scale is 2^4 in the lanes of group gs and 2^0 elsewhere.  C[0][0] = 4 * (16 if group gs scales those bytes else 1).
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mx_map.hip -o /tmp/mxmap && /tmp/mxmap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(float* out) {
    const int l = threadIdx.x, lg = l >> 4;
    const int g = blockIdx.x >> 3, d = blockIdx.x & 7, gs = blockIdx.y;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (lg == g && i == d) ? 0x38383838 : 0;
        b[i] = 0x38383838;
    }
    const int sa = lg == gs ? 0x83 : 0x7f, sb = 0x7f;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    out[(blockIdx.y * 32 + blockIdx.x) * 64 + l] = acc[0];  // every lane stores: the MFMA must not end up under a lane mask
}

int main() {
    float* d; hipMalloc(&d, 128 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(32, 4), dim3(64), 0, 0, d);
    static float hh[128 * 64]; hipMemcpy(hh, d, sizeof hh, hipMemcpyDeviceToHost);
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = hh[i * 64];
    printf("rows: data (lane group g, dword d); columns: scale lane group gs; entry = C[0][0]\n");
    for (int x = 0; x < 32; ++x) {
        printf("g=%d d=%d :", x >> 3, x & 7);
        for (int gs = 0; gs < 4; ++gs) printf(" %6.0f", h[gs * 32 + x]);
        printf("\n");
    }
    return 0;
}
