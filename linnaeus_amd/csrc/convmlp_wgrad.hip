// Fused weight gradients of the ConvNeXt pointwise MLP for gfx950 (bf16 storage):
//
//   dW1[4C, C] += dH^T . ln      db1[4C] += colsum(dH)      (pwconv1, blocks/convnext.py:60-61 backward)
//   dW2[C, 4C] += dz^T . act     db2[C]  += colsum(dz)      (pwconv2, blocks/convnext.py:64    backward)
//   with  act = GELU(ln . W1^T + b1),  dH = (dz . W2) * GELU'(ln . W1^T + b1)  RECOMPUTED per 32-row tile.
//
// Why: round 1 had lnx_convmlp_bwd write act and dH ([M, 4C] bf16 each: 2 x 616 MB at stage 0) only for two
// weight-gradient GEMMs to read them back -- half of the conv stages' backward HBM traffic.  Recomputing the hidden
// tile costs K = C MFMA products, which at C = 96/192 is far cheaper than 16 B/element of HBM round trip.
//
// Structure.  Orientation matters: the data-side kernel (convmlp.hip) splits ROWS over waves and keeps the hidden
// index in registers, so a row reduction there would need every wave to hold all of dW.  Here the HIDDEN index is split
// over the 8 waves instead (each wave owns 16*JT hidden units and therefore a fixed [16 JT, C] slice of the weight
// gradient in accumulator registers for the whole kernel), all waves walk the same 32-row tiles, and the products are
// oriented so that the hidden tile comes out of the MFMA as  D[row m][col hidden]: its registers are then directly the
// A operand (k = m) of the weight-gradient product -- no LDS transpose of act / dH (cdna_hip_programming.md "An
// accumulator tile as the next MFMA's operand": the k order inside the step is the permutation {4q+r, 16+4q+r}, and
// the other operand is fetched with the same permutation by two ds_read_b64_tr_b16 per fragment from the row-major
// ln / dz tile).
//
// One launch per matrix (WHICH = 1: dW1 + db1, WHICH = 2: dW2 + db2).  Both gradients of a [384, 96] slab are 295 KB of
// fp32 accumulators -- 144 of a wave's 256 registers -- and with what is left the loop could keep only one LDS read in
// flight per wait: measured 500+ us per launch, four times its MFMA time.  One matrix at a time leaves room to keep the
// row fragments of the whole tile resident and a hidden tile's weight fragments in flight under the previous tile's
// GELU; the price is computing h twice (5 instead of 4 products per tile).
//
//   LDS: W1 (and, for WHICH = 1, W2^T) slab image in the convmlp.hip "n-major" swizzled layout, filled once by
//        LDS-DMA | ln tile | dz tile ([32][2C + 16 B]) | b1 slab
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

// native vector types for everything an asm statement produces: "+v" (CW_PIN) cannot tie HIP's struct uint4 / uint2
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mfma16(f32x4_t& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

template <int NK, int JT, int WHICH> struct Gw {
    static constexpr int C = 32 * NK;
    static constexpr int CT = C / 16;
    static constexpr int NIMG = WHICH == 1 ? 2 : 1;  // dW1 needs W1 and W2^T, dW2 only W1
    static constexpr int HS = 128 * JT;              // hidden units of one slab (8 waves x JT tiles of 16)
    static constexpr int NP = HS / 64;               // 64-row parts of a slab image
    static constexpr int PART = NK * 4096;           // bytes of one part: [ks][64 rows][64 B]
    static constexpr int IMGS = NIMG * NP * PART;    // all images: parts interleaved [W1 part j | W2^T part j]
    static constexpr int PITCH = 2 * C + 16;         // bytes per tile row (+16: conflict-free b128 row reads)
    static constexpr int TILE = 32 * PITCH;
    static constexpr int LDS = IMGS + 2 * TILE + HS * 4;
    static constexpr int VEC = WHICH == 1 ? HS : C;  // bias-gradient entries of one partial
    static constexpr int SLAB = HS * C + VEC;        // floats of one partial: matrix slab [HS][C] | bias part
    static constexpr int NU = (4 * C + 511) / 512;   // 16-byte units per thread of one tile
};

// base VGPR + compile-time immediate offset (< 64 KiB): one address register serves every read of a family
#define CW_READ128(dst, base, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(OFF) : "memory")
#define CW_READTR(dst, base, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(OFF) : "memory")
// sched_barrier on both sides: VALU work placed before the wait (to run under the LDS latency) must not sink below it,
// and nothing that consumes the loaded registers may be hoisted above it
#define CW_WAITN(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// An asm load's destination counts as written at the asm statement: the compiler may copy it (tuple assembly, AGPR
// moves) BEFORE the data has landed.  CW_PIN, placed after the wait, makes the value opaque there, so every use --
// copies included -- is ordered behind the wait (cdna_hip_programming.md 5.7 item 1, form (ii)).
#define CW_PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ u32x2 pack4(const f32x4_t& v) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    const bf4 h = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return __builtin_bit_cast(u32x2, h);
}

// one hidden tile (both m-tiles = 4 element pairs in lockstep, common.hpp "_n" forms):
// dH = dA * GELU'(h + b) packed to bf16; returns the fp32 sum of the eight values (bias gradient)
__device__ __forceinline__ float dgelu_pack2(const f32x4_t (&h)[2], const f32x4_t (&d)[2], float b, u32x2& p0, u32x2& p1) {
    f32x2_t v[4], a[4], g[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        v[2 * m] = f32x2_t{h[m][0] + b, h[m][1] + b};
        v[2 * m + 1] = f32x2_t{h[m][2] + b, h[m][3] + b};
    }
    gelu_lean_grad2_n<4>(v, a, g);
    const f32x4_t dh0 = f32x4_t{d[0][0] * g[0].x, d[0][1] * g[0].y, d[0][2] * g[1].x, d[0][3] * g[1].y};
    const f32x4_t dh1 = f32x4_t{d[1][0] * g[2].x, d[1][1] * g[2].y, d[1][2] * g[3].x, d[1][3] * g[3].y};
    p0 = pack4(dh0);
    p1 = pack4(dh1);
    return ((dh0[0] + dh0[1]) + (dh0[2] + dh0[3])) + ((dh1[0] + dh1[1]) + (dh1[2] + dh1[3]));
}
// act = GELU(h + b) packed to bf16
__device__ __forceinline__ void gelu_pack2(const f32x4_t (&h)[2], float b, u32x2& p0, u32x2& p1) {
    f32x2_t v[4], a[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        v[2 * m] = f32x2_t{h[m][0] + b, h[m][1] + b};
        v[2 * m + 1] = f32x2_t{h[m][2] + b, h[m][3] + b};
    }
    gelu_lean2_n<4>(v, a);
    p0 = pack4(f32x4_t{a[0].x, a[0].y, a[1].x, a[1].y});
    p1 = pack4(f32x4_t{a[2].x, a[2].y, a[3].x, a[3].y});
}

// Diagnostic build only (-DCW_STAMP, loaded through LNX_LIB_PATH by tools/stamp_wgrad.py): per-phase s_memtime sums of
// every wave, written over the head of the workgroup's partial slab (the results are then garbage by design).
#ifdef CW_STAMP
#define CW_T(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                     __builtin_amdgcn_sched_barrier(0); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define CW_T(i) do { } while (0)
#endif

template <int NK, int JT, int WHICH>
__global__ __launch_bounds__(512) void convmlp_wgrad_kernel(const CwP p) {
    using G = Gw<NK, JT, WHICH>;
    constexpr int C = G::C, CT = G::CT, HS = G::HS, PART = G::PART, NIMG = G::NIMG, IMGS = G::IMGS, PITCH = G::PITCH, TILE = G::TILE, NU = G::NU;
    constexpr bool D1 = WHICH == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wimg = smem;
    unsigned char* lnT = smem + IMGS;
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
    // slab images (rows hb .. hb+HS-1 of the [4C, C] matrices), 1 KiB LDS-DMA pieces over the 8 waves
    {
        constexpr int NINS = 4 * NK;
        for (int qq = wave; qq < NIMG * G::NP * NINS; qq += 8) {
            const int which = qq / (G::NP * NINS);
            const int rem = qq % (G::NP * NINS);
            const int j = rem / NINS, i = rem % NINS;
            const int ks = i >> 2, rb = i & 3;
            const int row = 16 * rb + (lane >> 2), u = lane & 3;
            const unsigned char* W = which ? p.w2t : p.w1;
            const unsigned char* src = W + ((int64_t)(hb + 64 * j + row) * C + ks * 32 + ((u ^ key4(row >> 3)) << 3)) * 2;
            unsigned char* dst = wimg + (NIMG * j + which) * PART + i * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }

    // per-lane constants
    uint32_t wbase[JT];   // LDS address of this lane's B-fragment row of W1 for hidden tile jt, k-step 0 (W2^T: + PART)
    constexpr int NDB = D1 ? JT : (CT + 7) / 8;
    float dbacc[NDB];     // bias-gradient partials: db1 of hidden tile jt / db2 of column tiles wave + 8 i
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int jl = wave * 16 * JT + 16 * jt + s;
        const int r64 = jl & 63;
        wbase[jt] = lds0 + (uint32_t)((jl >> 6) * NIMG * PART + r64 * 64 + ((q ^ key4(r64 >> 3)) << 4));
    }
#pragma unroll
    for (int i = 0; i < NDB; ++i) dbacc[i] = 0.f;
    // ln-tile addresses; the dz tile is the same + TILE (immediate)
    const uint32_t a_base = lds0 + IMGS + (uint32_t)(s * PITCH + q * 16);                        // A-fragment rows (b128): + 16 mt PITCH + 64 ks
    const uint32_t t_base = lds0 + IMGS + (uint32_t)((4 * q + (s >> 2)) * PITCH + 8 * (s & 3));  // transposed reads: + 32 ct (+ 16 PITCH)

    f32x4_t dw[JT][CT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dw[jt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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

    float b1v[JT];  // bias of this lane's hidden unit j0 + s of every tile (registers: the GELU then has no LDS dependency)
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) b1v[jt] = b1s[wave * 16 * JT + 16 * jt + s];

    // The second-dispatched half of the workgroup (waves 4-7 share SIMDs with waves 0-3) loses every VALU / MFMA
    // arbitration by age: measured, its phase 1 took 3300 cycles against 2080 for the older half, which then idled at the
    // barrier.  One static priority raise evens them out (cdna_hip_programming.md T5, static form).
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#ifdef CW_STAMP
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    for (int t = t_begin; t < t_end; ++t) {
        commit();
        CW_T(0);
        __syncthreads();
        CW_T(1);
        if (t + 1 < t_end) issue(t + 1);  // next tile's rows: a whole tile of arithmetic covers the HBM latency

        // ---- phase 1: the wave's hidden tiles, software-pipelined over jt: the MFMAs of tile jt are issued, then the
        //      weight fragments of tile jt+1 are requested, then the GELU of tile jt-1 runs on the VALU -- under the
        //      MFMAs and the LDS latency.  Result: the hidden tile (act or dH) for rows 16 mt + {4q..4q+3}, hidden
        //      j0 + s, packed to bf16 = the A operand of phase 2.
        u32x4 al[2][NK], az[D1 ? 2 : 1][NK];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                CW_READ128(al[mt][ks], a_base, mt * 16 * PITCH + ks * 64);
                if (D1) CW_READ128(az[mt][ks], a_base, TILE + mt * 16 * PITCH + ks * 64);
            }
        // dW2 (W1 fragments only) has the registers to request EVERY hidden tile's fragments at once: one LDS wait per
        // tile instead of one per hidden tile
        constexpr bool ALLW = !D1 && JT * NK <= 12;
        u32x4 wf[ALLW ? JT * NK : NK], vf[D1 ? NK : 1];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            CW_READ128(wf[ks], wbase[0], ks * 4096);
            if (D1) CW_READ128(vf[ks], wbase[0], PART + ks * 4096);
        }
        if (ALLW) {
#pragma unroll
            for (int jt = 1; jt < JT; ++jt)
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) CW_READ128(wf[jt * NK + ks], wbase[jt], ks * 4096);
        }
        u32x4 pf[JT];                   // {mt 0: .x .y, mt 1: .z .w}
        f32x4_t hh[2][2], dd[2][2];     // [jt parity][mt]
#pragma unroll
        for (int jt = 0; jt <= JT; ++jt) {
            if (jt < JT) {
                if (!ALLW || jt == 0) CW_WAITN(0);
                if (ALLW && jt == 0) {
#pragma unroll
                    for (int i = NK; i < JT * NK; ++i) CW_PIN(wf[i]);
                }
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    if (!ALLW || jt == 0) CW_PIN(wf[ks]);
                    if (D1) CW_PIN(vf[ks]);
                    if (jt == 0) {
                        CW_PIN(al[0][ks]);
                        CW_PIN(al[1][ks]);
                        if (D1) {
                            CW_PIN(az[0][ks]);
                            CW_PIN(az[1][ks]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4_t& h0 = hh[jt & 1][0];
                f32x4_t& h1 = hh[jt & 1][1];
                f32x4_t& d0 = dd[jt & 1][0];
                f32x4_t& d1 = dd[jt & 1][1];
                h0 = h1 = d0 = d1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    const u32x4& w = wf[ALLW ? jt * NK + ks : ks];
                    mfma16(h0, al[0][ks], w);  // h[m][j]  = sum_c ln[m][c] W1[j][c]
                    mfma16(h1, al[1][ks], w);
                    if (D1) {
                        mfma16(d0, az[0][ks], vf[ks]);  // dA[m][j] = sum_c dz[m][c] W2^T[j][c]
                        mfma16(d1, az[1][ks], vf[ks]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!ALLW && jt + 1 < JT) {
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) {
                        CW_READ128(wf[ks], wbase[(jt + 1) % JT], ks * 4096);
                        if (D1) CW_READ128(vf[ks], wbase[(jt + 1) % JT], PART + ks * 4096);
                    }
                }
            }
            if (jt > 0) {
                const int jp = jt - 1;
                const float bj = b1v[jp];
                u32x2 e0, e1;
                if (D1) dbacc[jp] += dgelu_pack2(hh[jp & 1], dd[jp & 1], bj, e0, e1);
                else gelu_pack2(hh[jp & 1], bj, e0, e1);
                pf[jp] = u32x4{e0.x, e0.y, e1.x, e1.y};
            }
        }
        CW_T(2);
        // ---- phase 2: weight-gradient products, k = the tile's 32 rows in the order {4q+r, 16+4q+r}.  B operand: the
        //      transposed fragments of ln (dW1) or dz (dW2); column tile ct+1 is requested before the JT MFMAs of ct ----
        constexpr int TOFF = D1 ? 0 : TILE;
        u32x2 b0[2], b1r[2];
        CW_READTR(b0[0], t_base, TOFF);
        CW_READTR(b1r[0], t_base, TOFF + 16 * PITCH);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int cur = ct & 1, nxt = cur ^ 1;
            if (ct + 1 < CT) {
                CW_READTR(b0[nxt], t_base, TOFF + 32 * (ct + 1));
                CW_READTR(b1r[nxt], t_base, TOFF + 32 * (ct + 1) + 16 * PITCH);
                CW_WAITN(2);
            } else {
                CW_WAITN(0);
            }
            CW_PIN(b0[cur]);
            CW_PIN(b1r[cur]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 bf = u32x4{b0[cur].x, b0[cur].y, b1r[cur].x, b1r[cur].y};
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) mfma16(dw[jt][ct], pf[jt], bf);  // dW1[j][c] += dH^T ln   /   dW2^T[j][c] += act^T dz
            if (!D1 && slab == 0 && (ct & 7) == wave) {  // db2[c0 + s] += this lane's 8 rows of dz (wave-uniform branch)
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) a += __uint_as_float(bf[i] << 16) + __uint_as_float(bf[i] & 0xFFFF0000u);
                dbacc[ct >> 3] += a;
            }
        }
        CW_T(3);
        __syncthreads();  // every wave is done with this tile's LDS image
        CW_T(4);
    }

    // ---- partial slab of this workgroup -> workspace ----
    float* slabp = p.ws + (int64_t)(slab * p.nsplit + split) * G::SLAB;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int jl0 = wave * 16 * JT + 16 * jt + 4 * q;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) slabp[(jl0 + r) * C + 16 * ct + s] = dw[jt][ct][r];
        if (D1) {
            float v = dbacc[jt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) slabp[HS * C + wave * 16 * JT + 16 * jt + s] = v;
        }
    }
#ifdef CW_STAMP
    if (lane == 0)
        for (int i = 0; i < 6; ++i) slabp[wave * 8 + i] = (float)tsum[i];
#endif
    if (!D1 && slab == 0) {
#pragma unroll
        for (int i = 0; i < (CT + 7) / 8; ++i) {
            const int ct = wave + 8 * i;
            float v = dbacc[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (ct < CT && q == 0) slabp[HS * C + 16 * ct + s] = v;
        }
    }
}

// gradients += sum over row ranges of the partial slabs, in a fixed order.  256-thread block = 32 float4 element groups
// x 8 lanes that each sum every 8th row range (4 independent loads in flight per lane), then a shuffle tree over the 8
// lanes (a serial loop over up to 256 slabs per thread would be latency-bound).
//   which = 1: dst[j][c] += (dW1, [4C, C]);   which = 2: dst[c][j] += (dW2 in torch layout [C, 4C], slabs hold dW2^T)
__global__ __launch_bounds__(256) void convmlp_wgrad_reduce_kernel(const float* __restrict__ ws, int C, int HS, int nsplit, int slab_floats, int which,
                                                                   float* __restrict__ dmat, float* __restrict__ dvec) {
    const int H4 = 4 * C;
    const int nmat = H4 * C / 4;                   // float4 groups of the matrix
    const int nvec = (which == 1 ? H4 : C) / 4;    // float4 groups of the bias gradient
    const int g = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int sl = threadIdx.x & 7;
    if (g >= nmat + nvec) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add4 = [](float4& x, const float4& y) { x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; };
    auto lanesum = [](float4& x) {
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            x.x += __shfl_xor(x.x, o, 64); x.y += __shfl_xor(x.y, o, 64); x.z += __shfl_xor(x.z, o, 64); x.w += __shfl_xor(x.w, o, 64);
        }
    };
    const float* base;
    int j = 0, c = 0, k = 0;
    if (g < nmat) {
        j = (g * 4) / C;
        c = (g * 4) % C;
        base = ws + (int64_t)(j / HS) * nsplit * slab_floats + (j % HS) * C + c;
    } else {
        k = (g - nmat) * 4;
        base = which == 1 ? ws + (int64_t)(k / HS) * nsplit * slab_floats + HS * C + (k % HS) : ws + HS * C + k;  // db2: slab 0 only
    }
    int sp = sl;
    for (; sp + 24 < nsplit; sp += 32) {
        float4 u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const float4*>(base + (int64_t)(sp + 8 * i) * slab_floats);
#pragma unroll
        for (int i = 0; i < 4; ++i) add4(a, u[i]);
    }
    for (; sp < nsplit; sp += 8) add4(a, *reinterpret_cast<const float4*>(base + (int64_t)sp * slab_floats));
    lanesum(a);
    if (sl != 0) return;
    if (g < nmat) {
        if (which == 1) {
            float4* d = reinterpret_cast<float4*>(dmat + (int64_t)j * C + c);
            float4 o = *d;
            add4(o, a);
            *d = o;
        } else {
            dmat[(int64_t)(c + 0) * H4 + j] += a.x;
            dmat[(int64_t)(c + 1) * H4 + j] += a.y;
            dmat[(int64_t)(c + 2) * H4 + j] += a.z;
            dmat[(int64_t)(c + 3) * H4 + j] += a.w;
        }
    } else {
        float4* d = reinterpret_cast<float4*>(dvec + k);
        float4 o = *d;
        add4(o, a);
        *d = o;
    }
}

struct Shape {
    int HS, slab_floats, nslab;
};
// hidden tiles of 16 per wave (8 waves) for each matrix: {JT for dW1, JT for dW2}
inline bool jt_of(int C, int* j1, int* j2) {
    switch (C) {
        case 32: *j1 = 1; *j2 = 1; return true;
        case 64: *j1 = 2; *j2 = 2; return true;
        case 96: *j1 = 3; *j2 = 3; return true;
        case 128: *j1 = 2; *j2 = 2; return true;
        case 192: *j1 = 1; *j2 = 2; return true;  // dW1 needs both weight images in LDS: 128 hidden units per slab
        default: return false;
    }
}
inline Shape shape_of(int C, int jt, int which) {
    Shape sh;
    sh.HS = 128 * jt;
    sh.slab_floats = sh.HS * C + (which == 1 ? sh.HS : C);
    sh.nslab = 4 * C / sh.HS;
    return sh;
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
inline void split_of(int M, const Shape& sh, int* nsplit, int* tps, int* ntile, int* xcd_map) {
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

template <int NK, int JT, int WHICH>
void launch(const CwP& p, int grid, hipStream_t st) {
    static bool attr = false;
    constexpr int lds = Gw<NK, JT, WHICH>::LDS;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convmlp_wgrad_kernel<NK, JT, WHICH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((convmlp_wgrad_kernel<NK, JT, WHICH>), dim3(grid), dim3(512), lds, st, p);
}

}  // namespace

extern "C" int64_t lnx_convmlp_wgrad_ws_floats(int C, int M) {
    int j1, j2;
    if (!jt_of(C, &j1, &j2) || M <= 0) return 0;
    int64_t need = 0;
    for (int which = 1; which <= 2; ++which) {
        const Shape sh = shape_of(C, which == 1 ? j1 : j2, which);
        int nsplit, tps, ntile, xm;
        split_of(M, sh, &nsplit, &tps, &ntile, &xm);
        const int64_t n = (int64_t)sh.nslab * nsplit * sh.slab_floats;
        if (n > need) need = n;
    }
    return need;
}

extern "C" int lnx_convmlp_wgrad(const lnx_convmlp_wgrad_args* a, void* stream) {
    LNX_CHECK(a && a->ln && a->dz && a->w1 && a->w2t && a->b1 && a->dw1 && a->db1 && a->dw2 && a->db2 && a->ws, "lnx_convmlp_wgrad: null operand");
    int j1, j2;
    LNX_CHECK(a->dtype == LNX_BF16 && jt_of(a->C, &j1, &j2), "lnx_convmlp_wgrad: unsupported dtype %d / C %d (bf16, C in {32,64,96,128,192})", a->dtype, a->C);
    LNX_CHECK(a->M > 0, "lnx_convmlp_wgrad: empty");
    LNX_CHECK(a->ws_floats >= lnx_convmlp_wgrad_ws_floats(a->C, a->M), "lnx_convmlp_wgrad: workspace too small (%lld floats, need %lld)",
              (long long)a->ws_floats, (long long)lnx_convmlp_wgrad_ws_floats(a->C, a->M));
    hipStream_t st = (hipStream_t)stream;
    for (int which = 1; which <= 2; ++which) {  // the same workspace serves both launches (stream order)
        const Shape sh = shape_of(a->C, which == 1 ? j1 : j2, which);
        CwP p{};
        p.ln = (const unsigned char*)a->ln; p.dz = (const unsigned char*)a->dz; p.w1 = (const unsigned char*)a->w1; p.w2t = (const unsigned char*)a->w2t;
        p.b1 = a->b1; p.ws = a->ws; p.M = a->M; p.C = a->C;
        split_of(a->M, sh, &p.nsplit, &p.tps, &p.ntile, &p.xcd_map);
        const int grid = sh.nslab * p.nsplit;
        switch (a->C * 10 + which) {
            case 321: launch<1, 1, 1>(p, grid, st); break;
            case 322: launch<1, 1, 2>(p, grid, st); break;
            case 641: launch<2, 2, 1>(p, grid, st); break;
            case 642: launch<2, 2, 2>(p, grid, st); break;
            case 961: launch<3, 3, 1>(p, grid, st); break;
            case 962: launch<3, 3, 2>(p, grid, st); break;
            case 1281: launch<4, 2, 1>(p, grid, st); break;
            case 1282: launch<4, 2, 2>(p, grid, st); break;
            case 1921: launch<6, 1, 1>(p, grid, st); break;
            case 1922: launch<6, 2, 2>(p, grid, st); break;
        }
        LNX_LAUNCH_CHECK();
        const int groups = 4 * a->C * a->C / 4 + (which == 1 ? 4 * a->C : a->C) / 4;  // float4 element groups, 32 per block
        hipLaunchKernelGGL(convmlp_wgrad_reduce_kernel, dim3(cdiv(groups, 32)), dim3(256), 0, st, p.ws, a->C, sh.HS, p.nsplit, sh.slab_floats, which,
                           which == 1 ? a->dw1 : a->dw2, which == 1 ? a->db1 : a->db2);
        LNX_LAUNCH_CHECK();
    }
    return 0;
}
