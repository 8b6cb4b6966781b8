// Which per-lane access pattern of a GEMM epilogue can the memory pipeline sustain?  gfx950, bf16 [M, N] tensor,
// 256x128 workgroup tiles, each of 8 waves a 64x64 sub-tile, 16 bytes per lane per instruction in every pattern:
//   A  the MFMA-natural pattern of gemm_epilogue_fast: lane (s, g) owns row (s>>2)*16 + (s&3) + 4 mi, columns 16 g .. 16 g + 15
//      (adjacent lanes = different rows; an instruction touches 16 rows x 4 pieces of 16 B at a 32-B stride)
//   B  row-contiguous: instruction i covers rows 8 i .. 8 i + 7, 8 lanes x 16 B = the row's 128 B
//   C  quad-contiguous: 4 adjacent lanes = 64 contiguous bytes, 16 rows per instruction
//   D  the rows of A, but the 4 lanes that share a row (lanes s, s+16, s+32, s+48 -- NOT adjacent) write 64 contiguous bytes per
//      instruction: what a 4x4 exchange of 16-byte pieces between those lanes (v_permlane16/32_swap) turns pattern A into
// mode 0 = stores, 1 = loads (summed into a dummy), 2 = load + store (copy in place).
// Build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int PAT, int MODE>
__global__ __launch_bounds__(512) void k(unsigned char* buf, int M, int N, int tiles_n, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * 256 + wm * 64, n0 = tn * 128 + wn * 64;
    const int s = lane & 15, g = lane >> 4;
    uint4 v[8];
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int row, colb;  // row within the sub-tile, byte column within its 128-byte row
        if (PAT == 0) {
            row = (s >> 2) * 16 + (s & 3) + 4 * (i >> 1);
            colb = g * 32 + (i & 1) * 16;
        } else if (PAT == 1) {
            row = 8 * i + (lane >> 3);
            colb = (lane & 7) * 16;
        } else if (PAT == 2) {
            row = (lane >> 2) + 16 * (i >> 1);
            colb = (lane & 3) * 16 + (i & 1) * 64;
        } else {
            row = (s >> 2) * 16 + (s & 3) + 4 * (i >> 1);
            colb = g * 16 + (i & 1) * 64;
        }
        int m = m0 + row;
        if (m >= M) m = M - 1;
        unsigned char* p = buf + ((size_t)m * N + n0) * 2 + colb;
        if (MODE >= 1) v[i] = *reinterpret_cast<const uint4*>(p);
        else v[i] = make_uint4(lane, i, row, colb);
        if (MODE == 1) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int row, colb;
            if (PAT == 0) {
                row = (s >> 2) * 16 + (s & 3) + 4 * (i >> 1);
                colb = g * 32 + (i & 1) * 16;
            } else if (PAT == 1) {
                row = 8 * i + (lane >> 3);
                colb = (lane & 7) * 16;
            } else if (PAT == 2) {
                row = (lane >> 2) + 16 * (i >> 1);
                colb = (lane & 3) * 16 + (i & 1) * 64;
            } else {
                row = (s >> 2) * 16 + (s & 3) + 4 * (i >> 1);
                colb = g * 16 + (i & 1) * 64;
            }
            const int m = m0 + row;
            if (m < M) *reinterpret_cast<uint4*>(buf + ((size_t)m * N + n0) * 2 + colb) = v[i];
        }
    } else if (acc == 0x12345678u) {
        sink[0] = acc;
    }
}

template <int PAT, int MODE>
static void run(unsigned char* buf, int M, int N, unsigned* sink) {
    const int tiles_n = N / 128, tiles_m = (M + 255) / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<PAT, MODE>), dim3(tiles_m * tiles_n), dim3(512), 0, 0, buf, M, N, tiles_n, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    const double bytes = (double)M * N * 2 * (MODE == 2 ? 2 : 1);
    printf("pattern %c mode %s  M=%d N=%d: %7.1f us  %5.2f TB/s\n", "ABCD"[PAT], MODE == 0 ? "store" : MODE == 1 ? "load " : "copy ", M, N, best * 1e3,
           bytes / (best * 1e-3) / 1e12);
}

int main() {
    const int M = 50944;
    unsigned char* buf;
    unsigned* sink;
    hipMalloc(&buf, (size_t)M * 3072 * 2);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, (size_t)M * 3072 * 2);
    for (int N : {384, 768, 1536}) {
        run<0, 0>(buf, M, N, sink);
        run<1, 0>(buf, M, N, sink);
        run<2, 0>(buf, M, N, sink);
        run<3, 0>(buf, M, N, sink);
        run<0, 1>(buf, M, N, sink);
        run<1, 1>(buf, M, N, sink);
        run<2, 1>(buf, M, N, sink);
        run<3, 1>(buf, M, N, sink);
        run<0, 2>(buf, M, N, sink);
        run<1, 2>(buf, M, N, sink);
        run<2, 2>(buf, M, N, sink);
        run<3, 2>(buf, M, N, sink);
    }
    return 0;
}
