// Registers the measurement kernels of tools/experiments/ with the product's NT dispatcher (gemm2.hip: g_nt_experiment).
// LNX_NT_V8=1 / LNX_NT_V5=1 (read per launch, so one process can compare) route every product the kernel can run through it.
#include <cstdlib>

#include "gemm_common.hpp"

namespace lnxg {
bool nt_v8_ok(const GemmP& p, int f, bool out_f32);
int launch_nt_v8(const GemmP& p, int f, bool out_f32, hipStream_t st);
bool nt_v5_ok(const GemmP& p, int f);
int launch_nt_v5(const GemmP& p, int f, bool out_f32, hipStream_t st);

static int experiment_dispatch(const GemmP& p, int f, bool out_f32, hipStream_t st) {
    const char* e8 = getenv("LNX_NT_V8");
    if (e8 && atoi(e8) == 1 && nt_v8_ok(p, f, out_f32)) {
        note_nt_kernel(LNX_NT_KERNEL_EXPERIMENT);
        return launch_nt_v8(p, f, out_f32, st);
    }
    const char* e5 = getenv("LNX_NT_V5");
    if (e5 && atoi(e5) == 1 && nt_v5_ok(p, f)) {
        note_nt_kernel(LNX_NT_KERNEL_EXPERIMENT);
        return launch_nt_v5(p, f, out_f32, st);
    }
    return 1;  // declined: the product's own choice runs
}

static const bool registered = (g_nt_experiment = &experiment_dispatch, true);
}  // namespace lnxg
