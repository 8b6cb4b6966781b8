// Internal: MFMA variants of the depthwise 7x7 kernels (dwconv_mfma.hip), dispatched by the C-ABI entry points in
// dwconv.hip when an operand is bf16 (the production compute type).  Return value as the entry points.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/lnx.h"

bool lnx_dwconv_mfma_enabled();
int lnx_dwconv7_mfma_fwd(const lnx_dwconv_args* a, hipStream_t st);
int lnx_dwconv7_mfma_wgrad(const lnx_dwconv_wgrad_args* a, hipStream_t st);
