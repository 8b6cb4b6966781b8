// Soft-label cross entropy (forward + logits gradient in one pass) for gfx950.
// Reference: TaxonomyAwareLabelSmoothingCE.forward (loss/taxonomy_label_smoothing.py:233-405):
//   log_probs = log_softmax(logits); per_sample = -sum_c soft_labels[target, c] * log_probs[c];
//   per_sample = 0 where target == ignore_index; per_sample *= class_weight[target] (optional)
// and, with soft == NULL, F.cross_entropy(logits, target, label_smoothing = eps) per sample.
// One 256-thread workgroup per sample row; everything fp32; exp/log only on the row maximum-shifted values.
#include "common.hpp"
#include "../../include/lnx.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // red may still be read from the previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

__device__ __forceinline__ void softce_row(const lnx_softce_args& a, const int b) {
    __shared__ float red[4];
    const float* x = a.logits + (int64_t)b * a.ld;
    const int64_t t = a.target[b];
    const bool bad = t < 0 || t >= a.C;  // out-of-range targets produce NaN loss and zero gradient (the host checks them when asked to)
    const bool ignored = a.ignore_index >= 0 && t == a.ignore_index;
    const float* srow = (a.soft != nullptr && !bad) ? a.soft + t * (int64_t)a.C : nullptr;
    const float on = 1.0f - a.smoothing, off = a.smoothing / (float)a.C;

    float mx = -INFINITY;
    for (int c = threadIdx.x; c < a.C; c += 256) mx = fmaxf(mx, x[c]);
    mx = block_reduce(mx, red, true);
    float se = 0.f, sx = 0.f, ss = 0.f;  // sum exp, sum S*x, sum S
    for (int c = threadIdx.x; c < a.C; c += 256) {
        const float v = x[c];
        se += __expf(v - mx);
        const float sv = srow ? srow[c] : (c == t ? on + off : off);
        sx = fmaf(sv, v, sx);
        ss += sv;
    }
    se = block_reduce(se, red, false);
    sx = block_reduce(sx, red, false);
    ss = block_reduce(ss, red, false);
    const float lse = mx + __logf(se);
    const float cw = (a.class_weight != nullptr && !bad) ? a.class_weight[t] : 1.0f;
    float loss = bad ? NAN : (ignored ? 0.f : cw * (lse * ss - sx));
    const float g = (ignored || bad) ? 0.f : a.scale * (a.row_scale ? a.row_scale[b] : 1.0f) * cw;
    if (threadIdx.x == 0) {
        if (a.loss) a.loss[b] = loss;
        if (a.loss_sum) atomicAdd(a.loss_sum, a.scale * (a.row_scale ? a.row_scale[b] : 1.0f) * loss);
    }
    if (a.dlogits) {
        float* d = a.dlogits + (int64_t)b * a.ldd;
        const float inv = 1.0f / se;
        for (int c = threadIdx.x; c < a.C; c += 256) {
            const float p = __expf(x[c] - mx) * inv;
            const float sv = srow ? srow[c] : (c == t ? on + off : off);
            d[c] = g * (p * ss - sv);
        }
    }
}

__global__ __launch_bounds__(256) void softce_kernel(const lnx_softce_args a) { softce_row(a, blockIdx.x); }

// all tasks of a multi-task criterion in one launch: blockIdx.y = task (the arguments travel in the kernel-argument segment)
struct SoftceMulti {
    lnx_softce_args a[LNX_SOFTCE_MAX_TASKS];
};
__global__ __launch_bounds__(256) void softce_multi_kernel(const SoftceMulti m) {
    const lnx_softce_args& a = m.a[blockIdx.y];
    if ((int)blockIdx.x < a.B) softce_row(a, blockIdx.x);
}

}  // namespace

extern "C" int lnx_softce_multi(const lnx_softce_args* args, int n, void* stream) {
    LNX_CHECK(args && n > 0 && n <= LNX_SOFTCE_MAX_TASKS, "lnx_softce_multi: 1..%d argument sets, got %d", LNX_SOFTCE_MAX_TASKS, n);
    SoftceMulti m;
    int bmax = 0;
    for (int i = 0; i < n; ++i) {
        const lnx_softce_args* a = args + i;
        LNX_CHECK(a->logits && a->target, "lnx_softce_multi: null operand in set %d", i);
        LNX_CHECK(a->B > 0 && a->C > 0 && a->ld >= a->C, "lnx_softce_multi: bad shape B=%d C=%d ld=%lld in set %d", a->B, a->C, (long long)a->ld, i);
        LNX_CHECK(a->dlogits == nullptr || a->ldd >= a->C, "lnx_softce_multi: ldd < C in set %d", i);
        LNX_CHECK(a->smoothing >= 0.f && a->smoothing < 1.f, "lnx_softce_multi: smoothing must be in [0, 1)");
        m.a[i] = *a;
        bmax = a->B > bmax ? a->B : bmax;
    }
    hipLaunchKernelGGL(softce_multi_kernel, dim3(bmax, n), dim3(256), 0, (hipStream_t)stream, m);
    LNX_LAUNCH_CHECK();
    return 0;
}

extern "C" int lnx_softce(const lnx_softce_args* a, void* stream) {
    LNX_CHECK(a && a->logits && a->target, "lnx_softce: null operand");
    LNX_CHECK(a->B > 0 && a->C > 0 && a->ld >= a->C, "lnx_softce: bad shape B=%d C=%d ld=%lld", a->B, a->C, (long long)a->ld);
    LNX_CHECK(a->dlogits == nullptr || a->ldd >= a->C, "lnx_softce: ldd < C");
    LNX_CHECK(a->smoothing >= 0.f && a->smoothing < 1.f, "lnx_softce: smoothing must be in [0, 1)");
    hipLaunchKernelGGL(softce_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a);
    LNX_LAUNCH_CHECK();
    return 0;
}
