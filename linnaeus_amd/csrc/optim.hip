// Multi-tensor AdamW with a fused global-norm clip for gfx950 ("optimizer + step glue").
// Replaces, for the model's parameters: torch.optim.AdamW.step (decoupled weight decay, bias correction) as built
// by linnaeus/optimizers/build.py, and the gradient-norm / clip passes of train.py:282-308
// (clip_grad_norm_: coef = min(1, max_norm / (total_norm + 1e-6)), gradients scaled by coef).
// Launches per step over a device descriptor table (one entry per parameter tensor): sum of squares of all gradients
// (partials per workgroup + a fixed-order fold; only when clipping), then the update, which reads the clip coefficient
// from that scalar.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

constexpr int OPT_ELEMS = 4096;  // elements per workgroup

__device__ __forceinline__ const lnx_adamw_desc& find_desc(const lnx_adamw_desc* __restrict__ descs, int ndesc, int block) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_start <= block) lo = mid;
        else hi = mid - 1;
    }
    return descs[lo];
}

// one partial per workgroup, folded in a fixed order by grad_sumsq_fold_kernel: the same gradients give the same bits on every
// rank and every run.  (A float atomicAdd per workgroup did not: the clip coefficient then differs in its last bits between
// data-parallel ranks that hold identical all-reduced gradients, and their parameters drift apart step by step.)
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const lnx_adamw_desc* __restrict__ descs, int ndesc, float* __restrict__ part) {
    __shared__ float red[4];
    const lnx_adamw_desc& d = find_desc(descs, ndesc, blockIdx.x);
    const int64_t base = (int64_t)(blockIdx.x - d.block_start) * OPT_ELEMS;
    const int64_t end = min(d.n, base + OPT_ELEMS);
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(d.g) & 15) == 0) {
        for (int64_t i = base + 4 * threadIdx.x; i + 3 < end; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(d.g + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const int64_t tail = base + ((end - base) & ~(int64_t)3);
        if (tail + threadIdx.x < end) {
            const float v = d.g[tail + threadIdx.x];
            s += v * v;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += 256) s += d.g[i] * d.g[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void grad_sumsq_fold_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains per thread: loads in flight
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        s0 += part[i];
        s1 += part[i + 1024];
        s2 += part[i + 2048];
        s3 += part[i + 3072];
    }
    for (; i < n; i += 1024) s0 += part[i];
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k];
        *out = t;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(const lnx_adamw_desc* __restrict__ descs, int ndesc, const lnx_adamw_hyper h, const float* __restrict__ sumsq,
                                                    float max_norm) {
    const lnx_adamw_desc& d = find_desc(descs, ndesc, blockIdx.x);
    const int gi = d.group;
    const float lr = h.lr[gi], b1 = h.beta1[gi], b2 = h.beta2[gi], eps = h.eps[gi], wd = h.weight_decay[gi];
    const float omb1 = h.omb1[gi], omb2 = h.omb2[gi];
    const float step_size = lr / h.bias_c1[gi], inv_sqrt_bc2 = 1.0f / sqrtf(h.bias_c2[gi]);
    float coef = 1.0f;
    if (sumsq != nullptr && max_norm > 0.f) coef = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    const int64_t base = (int64_t)(blockIdx.x - d.block_start) * OPT_ELEMS;
    const int64_t end = min(d.n, base + OPT_ELEMS);
    auto upd = [&](float g, float& p, float& m, float& v) __attribute__((always_inline)) {
        g *= coef;
        p *= 1.0f - lr * wd;
        m = fmaf(b1, m, omb1 * g);
        v = fmaf(b2, v, omb2 * g * g);
        const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
        p -= step_size * (m / denom);
    };
    // 16-byte accesses, the workgroup's four rounds of loads all in flight before the first use (at 4 bytes a lane the
    // 120 M-parameter xl step took 2.0 ms for 3.4 GB: 1.65 TB/s)
    const bool vec = ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g) | reinterpret_cast<uintptr_t>(d.m) | reinterpret_cast<uintptr_t>(d.v)) & 15) == 0;
    int64_t done = base;
    if (vec) {
        constexpr int R = OPT_ELEMS / 1024;
        float4 g4[R], p4[R], m4[R], v4[R];
        const int64_t full = base + ((end - base) & ~(int64_t)3);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = base + 1024 * r + 4 * threadIdx.x;
            const int64_t ic = i + 3 < full ? i : base;  // unconditional loads (a workgroup's first four elements always exist when full > base)
            if (full > base) {
                g4[r] = *reinterpret_cast<const float4*>(d.g + ic);
                p4[r] = *reinterpret_cast<const float4*>(d.p + ic);
                m4[r] = *reinterpret_cast<const float4*>(d.m + ic);
                v4[r] = *reinterpret_cast<const float4*>(d.v + ic);
            }
        }
        if (full > base) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i = base + 1024 * r + 4 * threadIdx.x;
                if (i + 3 < full) {
                    upd(g4[r].x, p4[r].x, m4[r].x, v4[r].x);
                    upd(g4[r].y, p4[r].y, m4[r].y, v4[r].y);
                    upd(g4[r].z, p4[r].z, m4[r].z, v4[r].z);
                    upd(g4[r].w, p4[r].w, m4[r].w, v4[r].w);
                    *reinterpret_cast<float4*>(d.p + i) = p4[r];
                    *reinterpret_cast<float4*>(d.m + i) = m4[r];
                    *reinterpret_cast<float4*>(d.v + i) = v4[r];
                }
            }
        }
        done = full;
    }
    for (int64_t i = done + threadIdx.x; i < end; i += 256) {
        float p = d.p[i], m = d.m[i], v = d.v[i];
        upd(d.g[i], p, m, v);
        d.p[i] = p;
        d.m[i] = m;
        d.v[i] = v;
    }
}

}  // namespace

extern "C" int lnx_adamw_blocks(int64_t numel) { return (int)((numel + OPT_ELEMS - 1) / OPT_ELEMS); }

extern "C" int lnx_grad_sumsq(const lnx_adamw_desc* descs_dev, int ndesc, int total_blocks, float* out, float* ws, void* stream) {
    LNX_CHECK(descs_dev && ndesc > 0 && total_blocks > 0 && out && ws, "lnx_grad_sumsq: bad arguments (ws: total_blocks floats)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(total_blocks), dim3(256), 0, st, descs_dev, ndesc, ws);
    hipLaunchKernelGGL(grad_sumsq_fold_kernel, dim3(1), dim3(1024), 0, st, ws, total_blocks, out);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_adamw_step(const lnx_adamw_desc* descs_dev, int ndesc, int total_blocks, const lnx_adamw_hyper* hyper, const float* sumsq, float max_norm,
                              void* stream) {
    LNX_CHECK(descs_dev && ndesc > 0 && total_blocks > 0 && hyper, "lnx_adamw_step: bad arguments");
    LNX_CHECK(hyper->ngroups > 0 && hyper->ngroups <= LNX_ADAMW_MAX_GROUPS, "lnx_adamw_step: ngroups=%d (max %d)", hyper->ngroups, LNX_ADAMW_MAX_GROUPS);
    for (int i = 0; i < hyper->ngroups; ++i)
        LNX_CHECK(hyper->bias_c1[i] > 0.f && hyper->bias_c2[i] > 0.f, "lnx_adamw_step: bias corrections of group %d must be positive (step >= 1)", i);
    hipLaunchKernelGGL(adamw_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, ndesc, *hyper, sumsq, max_norm);
    LNX_LAUNCH_CHECK();
    return 0;
}
