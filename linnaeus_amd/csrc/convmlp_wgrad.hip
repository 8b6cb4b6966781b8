// Fused weight gradients of the ConvNeXt pointwise MLP for gfx950 (bf16 storage):
//
//   dW1[4C, C] += dH^T . ln      db1[4C] += colsum(dH)      (pwconv1, blocks/convnext.py:60-61 backward)
//   dW2[C, 4C] += dz^T . act     db2[C]  += colsum(dz)      (pwconv2, blocks/convnext.py:64    backward)
//   with  act = GELU(ln . W1^T + b1),  dH = (dz . W2) * GELU'(ln . W1^T + b1)  RECOMPUTED per 32-row tile.
//
// Why: round 1 had lnx_convmlp_bwd write act and dH ([M, 4C] bf16 each: 2 x 616 MB at stage 0) only for two
// weight-gradient GEMMs to read them back -- half of the conv stages' backward HBM traffic.  Recomputing the hidden
// tile costs two K = C MFMA products, which at C = 96/192 is far cheaper than 16 B/element of HBM round trip.
//
// Structure.  Orientation matters: the data-side kernel (convmlp.hip) splits ROWS over waves and keeps the hidden
// index in registers, so a row reduction there would need every wave to hold all of dW (295 KB).  Here the HIDDEN
// index is split over the 8 waves instead (each wave owns 16*JT hidden units and therefore a fixed [16 JT, C] slice of
// dW1 and of dW2^T in accumulator registers for the whole kernel), all waves walk the same 32-row tiles, and the
// products are oriented so that the hidden tile comes out of the MFMA as  D[row m][col hidden]: its registers are then
// directly the A operand (k = m) of the two weight-gradient products -- no LDS transpose of act / dH
// (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand": the k order inside the step is the
// permutation {4q+r, 16+4q+r}, and the other operand is fetched with the same permutation by two
// ds_read_b64_tr_b16 per fragment from the row-major ln / dz tile).
//
//   LDS: W1 slab image | W2^T slab image (the convmlp.hip "n-major" swizzled layout, filled once by LDS-DMA) |
//        ln tile | dz tile ([32][2C + 16 B]: the +16 makes the b128 row reads conflict-free) | b1 slab
//   grid: nslab (= 4C / (128 JT) hidden slabs) x nsplit row ranges, <= one workgroup per CU; every workgroup stores
//        its partial slab to a workspace and a second kernel sums the row ranges in a fixed order (bit-reproducible,
//        no float atomics) into the fp32 gradients.
#include "common.hpp"
#include "../../include/lnx.h"


namespace {

struct CwP {
    const unsigned char* ln;
    const unsigned char* dz;
    const unsigned char* w1;
    const unsigned char* w2t;
    const float* b1;
    float* ws;
    int M, C, nsplit, tps, ntile;
    int xcd_map;  // 1: blocks b and b + 8 share an XCD (round-robin dispatch) -> give them the slabs of ONE row range (same ln / dz tiles in that L2)
};

__device__ __forceinline__ int key4(int a) { return (4 - (a & 3)) & 3; }

__device__ __forceinline__ void mfma16(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

template <int NK, int JT> struct Gw {
    static constexpr int C = 32 * NK;
    static constexpr int CT = C / 16;
    static constexpr int HS = 128 * JT;            // hidden units of one slab (8 waves x JT tiles of 16)
    static constexpr int NP = HS / 64;             // 64-row parts of a slab image
    static constexpr int PART = NK * 4096;         // bytes of one part: [ks][64 rows][64 B]
    static constexpr int IMG = NP * PART;          // == HS * C * 2
    static constexpr int PITCH = 2 * C + 16;       // bytes per tile row
    static constexpr int TILE = 32 * PITCH;
    static constexpr int LDS = 2 * IMG + 2 * TILE + HS * 4;
    static constexpr int SLAB = 2 * HS * C + HS + C;  // floats of one partial: dW1 slab | dW2^T slab | db1 | db2
    static constexpr int NU = (4 * C + 511) / 512;    // 16-byte units per thread of one tile
};

// base VGPR + compile-time immediate offset (< 64 KiB): one address register serves every read of a family
#define CW_READ128(dst, base, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(OFF) : "memory")
#define CW_READTR(dst, base, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(OFF) : "memory")
#define CW_WAIT() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ uint2 pack4(const f32x4_t& v) {
    uint2 r;
    bf16_t* h = reinterpret_cast<bf16_t*>(&r);
    h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
    return r;
}

template <int NK, int JT>
__global__ __launch_bounds__(512) void convmlp_wgrad_kernel(const CwP p) {
    using G = Gw<NK, JT>;
    constexpr int C = G::C, CT = G::CT, HS = G::HS, PART = G::PART, IMG = G::IMG, PITCH = G::PITCH, TILE = G::TILE, NU = G::NU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wimg = smem;  // parts interleaved [W1 part j | W2^T part j]: the W2^T fragment is the W1 address + PART (an immediate)
    unsigned char* lnT = smem + 2 * IMG;
    unsigned char* dzT = lnT + TILE;
    float* b1s = reinterpret_cast<float*>(dzT + TILE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q = lane >> 4;
    const int nslab = 4 * C / HS;
    int slab, split;
    if (p.xcd_map) {
        const int xg = blockIdx.x & 7, y = blockIdx.x >> 3;
        slab = y % nslab;
        split = xg + 8 * (y / nslab);
    } else {
        slab = blockIdx.x % nslab;
        split = blockIdx.x / nslab;
    }
    const int hb = slab * HS;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    for (int i = tid; i < HS; i += 512) b1s[i] = p.b1[hb + i];
    // slab images of W1 and W2^T (rows hb .. hb+HS-1 of the [4C, C] matrices), 1 KiB LDS-DMA pieces over the 8 waves
    {
        constexpr int NINS = 4 * NK;
        for (int qq = wave; qq < 2 * G::NP * NINS; qq += 8) {
            const int which = qq / (G::NP * NINS);
            const int rem = qq % (G::NP * NINS);
            const int j = rem / NINS, i = rem % NINS;
            const int ks = i >> 2, rb = i & 3;
            const int row = 16 * rb + (lane >> 2), u = lane & 3;
            const unsigned char* W = which ? p.w2t : p.w1;
            const unsigned char* src = W + ((int64_t)(hb + 64 * j + row) * C + ks * 32 + ((u ^ key4(row >> 3)) << 3)) * 2;
            unsigned char* dst = wimg + (2 * j + which) * PART + i * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }

    // per-lane constants
    uint32_t wbase[JT];   // LDS address of this lane's B-fragment row of W1 for hidden tile jt, k-step 0 (W2^T: + PART)
    float db1acc[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int jl = wave * 16 * JT + 16 * jt + s;
        const int r64 = jl & 63;
        wbase[jt] = lds0 + (uint32_t)((jl >> 6) * 2 * PART + r64 * 64 + ((q ^ key4(r64 >> 3)) << 4));
        db1acc[jt] = 0.f;
    }
    // ln-tile addresses; the dz tile is the same + TILE (immediate)
    const uint32_t a_base = lds0 + 2 * IMG + (uint32_t)(s * PITCH + q * 16);                        // A-fragment rows (b128): + 16 mt PITCH + 64 ks
    const uint32_t t_base = lds0 + 2 * IMG + (uint32_t)((4 * q + (s >> 2)) * PITCH + 8 * (s & 3));  // transposed reads: + 32 ct (+ 16 PITCH)

    f32x4_t dw1[JT][CT], dw2[JT][CT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            dw1[jt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dw2[jt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    constexpr int NDB = (CT + 7) / 8;  // db2 column tiles of this wave: ct = wave + 8 i  (one register each: VALU sums of the dz fragment)
    float db2acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) db2acc[i] = 0.f;

    const int t_begin = split * p.tps;
    const int t_end = min(p.ntile, t_begin + p.tps);

    // tile staging: thread -> 16-byte unit u of the [32][C] tile (row u / (C/8), unit u % (C/8)); rows >= M are zero
    uint4 pln[NU], pdz[NU];
    auto issue = [&](int t) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = tid + 512 * k;
            pln[k] = make_uint4(0u, 0u, 0u, 0u);
            pdz[k] = make_uint4(0u, 0u, 0u, 0u);
            if (u < 4 * C) {
                const int row = u / (C / 8), cu = u % (C / 8);
                const int m = t * 32 + row;
                if (m < p.M) {
                    const int64_t off = ((int64_t)m * C + cu * 8) * 2;
                    pln[k] = ld16(p.ln + off);
                    pdz[k] = ld16(p.dz + off);
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = tid + 512 * k;
            if (u < 4 * C) {
                const int row = u / (C / 8), cu = u % (C / 8);
                st16(lnT + row * PITCH + cu * 16, pln[k]);
                st16(dzT + row * PITCH + cu * 16, pdz[k]);
            }
        }
    };

    if (t_begin < t_end) issue(t_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight images (LDS-DMA) and the first tile's registers
    __syncthreads();                                  // b1s + images visible to every wave

    for (int t = t_begin; t < t_end; ++t) {
        commit();
        __syncthreads();

        // ---- hidden tile of this wave: act and dH for rows 16 mt + {4q..4q+3}, hidden j0 + s, packed to bf16 ----
        uint2 pa[JT][2], pd[JT][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            uint4 al[NK];
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) CW_READ128(al[ks], a_base, mt * 16 * PITCH + ks * 64);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4_t h = f32x4_t{0.f, 0.f, 0.f, 0.f}, da = f32x4_t{0.f, 0.f, 0.f, 0.f};
                // Register budget at C = 96 is what the 144 accumulator registers leave: only the ln fragments stay
                // resident over the hidden tiles; the dz and weight fragments are fetched per k-step (the W1 one a step
                // ahead) and the partner wave of the SIMD covers the LDS latency.
                uint4 wf[2], vf, az;
                CW_READ128(wf[0], wbase[jt], 0);
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    CW_READ128(az, a_base, TILE + mt * 16 * PITCH + ks * 64);
                    CW_READ128(vf, wbase[jt], PART + ks * 4096);
                    if (ks + 1 < NK) {
                        CW_READ128(wf[(ks + 1) & 1], wbase[jt], (ks + 1) * 4096);
                        asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma16(h, al[ks], wf[ks & 1]);   // h[m][j]  = sum_c ln[m][c] W1[j][c]
                    mfma16(da, az, vf);              // dA[m][j] = sum_c dz[m][c] W2^T[j][c]
                }
                const float bj = b1s[wave * 16 * JT + 16 * jt + s];
                // h = GELU(h + b1), da = dA * GELU'(h + b1); one element pair at a time (fewer live temporaries)
                f32x2_t a0, d0, a1, d1;
                gelu_lean_grad2(f32x2_t{h[0] + bj, h[1] + bj}, a0, d0);
                const f32x2_t dh0 = f32x2_t{da[0] * d0.x, da[1] * d0.y};
                __builtin_amdgcn_sched_barrier(0);
                gelu_lean_grad2(f32x2_t{h[2] + bj, h[3] + bj}, a1, d1);
                const f32x2_t dh1 = f32x2_t{da[2] * d1.x, da[3] * d1.y};
                db1acc[jt] += (dh0.x + dh0.y) + (dh1.x + dh1.y);
                pa[jt][mt] = pack4(f32x4_t{a0.x, a0.y, a1.x, a1.y});
                pd[jt][mt] = pack4(f32x4_t{dh0.x, dh0.y, dh1.x, dh1.y});
            }
        }
        if (t + 1 < t_end) issue(t + 1);  // next tile's rows: in flight under the weight-gradient products
        // ---- weight-gradient products: k = the tile's 32 rows in the order {4q+r, 16+4q+r} ----
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            uint2 z0, z1, l0, l1;
            CW_READTR(z0, t_base, TILE + 32 * ct);
            CW_READTR(z1, t_base, TILE + 32 * ct + 16 * PITCH);
            CW_READTR(l0, t_base, 32 * ct);
            CW_READTR(l1, t_base, 32 * ct + 16 * PITCH);
            CW_WAIT();
            const uint4 bz = make_uint4(z0.x, z0.y, z1.x, z1.y);
            const uint4 bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                mfma16(dw2[jt][ct], make_uint4(pa[jt][0].x, pa[jt][0].y, pa[jt][1].x, pa[jt][1].y), bz);  // dW2^T[j][c] += act^T dz
                mfma16(dw1[jt][ct], make_uint4(pd[jt][0].x, pd[jt][0].y, pd[jt][1].x, pd[jt][1].y), bl);  // dW1[j][c]   += dH^T ln
            }
            if (slab == 0 && (ct & 7) == wave) {  // db2[c0 + s] += this lane's 8 rows of dz (wave-uniform branch, one tile per wave)
                const uint32_t w[4] = {bz.x, bz.y, bz.z, bz.w};
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) a += __uint_as_float(w[i] << 16) + __uint_as_float(w[i] & 0xFFFF0000u);
                db2acc[ct >> 3] += a;
            }
        }
        __syncthreads();  // every wave is done with this tile's LDS image
    }

    // ---- partial slab of this workgroup -> workspace ----
    float* slabp = p.ws + (int64_t)(slab * p.nsplit + split) * G::SLAB;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int jl0 = wave * 16 * JT + 16 * jt + 4 * q;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                slabp[(jl0 + r) * C + 16 * ct + s] = dw1[jt][ct][r];
                slabp[HS * C + (jl0 + r) * C + 16 * ct + s] = dw2[jt][ct][r];
            }
        float v = db1acc[jt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) slabp[2 * HS * C + wave * 16 * JT + 16 * jt + s] = v;
    }
    if (slab == 0) {
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int ct = wave + 8 * i;
            float v = db2acc[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (ct < CT && q == 0) slabp[2 * HS * C + HS + 16 * ct + s] = v;
        }
    }
}

// gradients += sum over row ranges of the partial slabs, in a fixed order
__global__ __launch_bounds__(256) void convmlp_wgrad_reduce_kernel(const float* __restrict__ ws, int C, int HS, int nsplit, int slab_floats,
                                                                   float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    const int H4 = 4 * C;
    const int nmat = H4 * C / 4;  // float4 groups of dW1 (and of dW2^T)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nmat) {
        const int j = (i * 4) / C, c = (i * 4) % C;
        const int slab = j / HS, jl = j % HS;
        const float* base = ws + (int64_t)slab * nsplit * slab_floats + jl * C + c;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 u = *reinterpret_cast<const float4*>(base + (int64_t)sp * slab_floats);
            const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)sp * slab_floats + HS * C);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        float4* d1 = reinterpret_cast<float4*>(dw1 + (int64_t)j * C + c);
        float4 o = *d1;
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        *d1 = o;
        dw2[(int64_t)(c + 0) * H4 + j] += b.x;
        dw2[(int64_t)(c + 1) * H4 + j] += b.y;
        dw2[(int64_t)(c + 2) * H4 + j] += b.z;
        dw2[(int64_t)(c + 3) * H4 + j] += b.w;
        return;
    }
    const int k = i - nmat;
    if (k < H4) {
        const int slab = k / HS, jl = k % HS;
        const float* base = ws + (int64_t)slab * nsplit * slab_floats + 2 * HS * C + jl;
        float a = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) a += base[(int64_t)sp * slab_floats];
        db1[k] += a;
    } else if (k < H4 + C) {
        const int c = k - H4;
        const float* base = ws + 2 * HS * C + HS + c;  // slab 0 only
        float a = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) a += base[(int64_t)sp * slab_floats];
        db2[c] += a;
    }
}

struct Shape {
    int HS, slab_floats, nslab;
};
inline bool shape_of(int C, Shape* sh) {
    int jt;
    switch (C) {
        case 32: jt = 1; break;
        case 64: jt = 2; break;
        case 96: jt = 3; break;
        case 128: jt = 2; break;
        case 192: jt = 1; break;
        default: return false;
    }
    sh->HS = 128 * jt;
    sh->slab_floats = 2 * sh->HS * C + sh->HS + C;
    sh->nslab = 4 * C / sh->HS;
    return true;
}
inline int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        cus = lnx_device_cus();
        if (cus <= 0) cus = 256;  // no device visible (e.g. sizing a plan on a build host): MI355X
    }
    return cus;
}
// row ranges: at most one workgroup per CU; with several slabs the count is a multiple of 8 so the slabs of one row
// range can be placed on one XCD
inline void split_of(int C, int M, const Shape& sh, int* nsplit, int* tps, int* ntile, int* xcd_map) {
    *ntile = cdiv(M, 32);
    int ns = device_cus() / sh.nslab;
    if (ns < 1) ns = 1;
    if (ns > *ntile) ns = *ntile;
    *tps = cdiv(*ntile, ns);
    *nsplit = cdiv(*ntile, *tps);
    *xcd_map = 0;
    if (sh.nslab > 1 && *nsplit >= 8) {
        const int ns8 = *nsplit / 8 * 8;
        const int tps8 = cdiv(*ntile, ns8);
        if (cdiv(*ntile, tps8) == ns8) {  // every row range non-empty
            *nsplit = ns8;
            *tps = tps8;
            *xcd_map = 1;
        }
    }
}

template <int NK, int JT>
int launch(const CwP& p, int grid, hipStream_t st) {
    static bool attr = false;
    constexpr int lds = Gw<NK, JT>::LDS;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_wgrad_kernel<NK, JT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((convmlp_wgrad_kernel<NK, JT>), dim3(grid), dim3(512), lds, st, p);
    return 0;
}

}  // namespace

extern "C" int64_t lnx_convmlp_wgrad_ws_floats(int C, int M) {
    Shape sh;
    if (!shape_of(C, &sh) || M <= 0) return 0;
    int nsplit, tps, ntile, xm;
    split_of(C, M, sh, &nsplit, &tps, &ntile, &xm);
    return (int64_t)sh.nslab * nsplit * sh.slab_floats;
}

extern "C" int lnx_convmlp_wgrad(const lnx_convmlp_wgrad_args* a, void* stream) {
    LNX_CHECK(a && a->ln && a->dz && a->w1 && a->w2t && a->b1 && a->dw1 && a->db1 && a->dw2 && a->db2 && a->ws, "lnx_convmlp_wgrad: null operand");
    Shape sh;
    LNX_CHECK(a->dtype == LNX_BF16 && shape_of(a->C, &sh), "lnx_convmlp_wgrad: unsupported dtype %d / C %d (bf16, C in {32,64,96,128,192})", a->dtype, a->C);
    LNX_CHECK(a->M > 0, "lnx_convmlp_wgrad: empty");
    CwP p{};
    p.ln = (const unsigned char*)a->ln; p.dz = (const unsigned char*)a->dz; p.w1 = (const unsigned char*)a->w1; p.w2t = (const unsigned char*)a->w2t;
    p.b1 = a->b1; p.ws = a->ws; p.M = a->M; p.C = a->C;
    split_of(a->C, a->M, sh, &p.nsplit, &p.tps, &p.ntile, &p.xcd_map);
    LNX_CHECK(a->ws_floats >= (int64_t)sh.nslab * p.nsplit * sh.slab_floats, "lnx_convmlp_wgrad: workspace too small (%lld floats, need %lld)",
              (long long)a->ws_floats, (long long)sh.nslab * p.nsplit * sh.slab_floats);
    hipStream_t st = (hipStream_t)stream;
    const int grid = sh.nslab * p.nsplit;
    switch (a->C) {
        case 32: launch<1, 1>(p, grid, st); break;
        case 64: launch<2, 2>(p, grid, st); break;
        case 96: launch<3, 3>(p, grid, st); break;
        case 128: launch<4, 2>(p, grid, st); break;
        case 192: launch<6, 1>(p, grid, st); break;
    }
    LNX_LAUNCH_CHECK();
    const int total = 4 * a->C * a->C / 4 + 4 * a->C + a->C;
    hipLaunchKernelGGL(convmlp_wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, p.ws, a->C, sh.HS, p.nsplit, sh.slab_floats, a->dw1, a->db1,
                       a->dw2, a->db2);
    LNX_LAUNCH_CHECK();
    return 0;
}
