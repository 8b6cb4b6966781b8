/*
 * lnx.h -- C ABI of the MI355X-native mFormerV1 forward/backward path.
 *
 * This is the drop-in boundary: plain C symbols taking raw device
 * pointers, sizes and a hipStream_t (passed as void*).  No torch types, no allocation
 * inside (workspaces are passed in), no hidden synchronisation: every call only enqueues
 * work on the given stream.  Every function returns 0 on success; on failure a message
 * is available from lnx_last_error().
 *
 * Each entry replaces work the reference does through torch.nn leaf modules; the
 * reference file:line it stands for is cited next to it (paths under /root/reference/).
 *
 * dtype codes: LNX_F32 = 0 (strict-parity mode), LNX_BF16 = 1 (production).  `dtype`
 * always names the storage type T of activations / GEMM operands; accumulation,
 * statistics, the residual stream, parameters and gradients are fp32.
 */
#ifndef LNX_H
#define LNX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNX_F32 0
#define LNX_BF16 1

const char* lnx_last_error(void);
int lnx_version(void);
/* number of compute units of the current device (used to size grids) */
int lnx_device_cus(void);

/* Row map for token buffers that carry E extra rows per sample:
 * phys_row = m + (m / group) * pad + off   (group == 0: identity). */
typedef struct lnx_rowmap {
    int group, pad, off;
} lnx_rowmap;

/* ------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T)         (MFMA, fp32 accumulate)
 * Replaces nn.Linear / nn.Conv2d(k=s) forward and their data-gradients:
 *   pwconv1/pwconv2 blocks/convnext.py:60-64,79-81; qkv/proj rope_2d_mhsa.py:292-294;
 *   fc1/fc2 blocks/mlp.py:36-39; stem/downsample convs mFormerV1.py:146, convnext.py:110;
 *   heads heads/linear_head.py:27; meta heads mFormerV1.py:291-306.
 * -----------------------------------------------------------------------------------*/
enum { LNX_ACT_NONE = 0, LNX_ACT_GELU = 1, LNX_ACT_RELU = 2, LNX_ACT_GELU_BWD = 3, LNX_ACT_RELU_BWD = 4 };
enum { LNX_ADDR_PLAIN = 0, LNX_ADDR_PATCH2 = 1 };

typedef struct lnx_gemm_args {
    int dtype;           /* T of A, W, aux, c2 (and of C unless out_f32) */
    int M, N, K;         /* K must be a multiple of 16/sizeof(T) */
    const void* A;       /* [M, K] rows, leading dimension lda (elements) */
    int64_t lda;
    const void* W;       /* [N, K] rows (torch Linear layout), leading dimension ldw */
    int64_t ldw;
    void* C;             /* [M, N] */
    int64_t ldc;
    int out_f32;         /* 1: C is fp32, 0: C is T */
    /* A addressing: PLAIN, or PATCH2 = gather 2x2/stride-2 patches from an NHWC tensor
     * [B, Hin, Win, Cin] with k = (kh*2 + kw)*Cin + c  (then K == 4*Cin, M == B*Hin/2*Win/2) */
    int a_mode, Hin, Win, Cin;
    /* C addressing: PLAIN with row map, or PATCH2 = scatter into NHWC (data-gradient of
     * the 2x2 conv; then N == 4*Cin and Hin/Win/Cin describe the destination) */
    int c_mode;
    lnx_rowmap c_map;
    /* epilogue, in this order: v = acc + bias[n]; c2 = v; act; v *= gamma[n];
     * v *= rowscale[m / rows_per_sample]; v += res[row, n]; C = v */
    const float* bias;     /* [N] or NULL */
    void* c2;              /* optional second output [M, N] of T, leading dimension ldc2 */
    int64_t ldc2;
    int act;
    const void* aux;       /* GELU_BWD: pre-activation, RELU_BWD: activation output; T */
    int64_t ldaux;
    const float* gamma;    /* [N] or NULL (LayerScale, blocks/convnext.py:82-83) */
    const float* rowscale; /* per-sample DropPath multiplier or NULL (drop_path.py:29-33) */
    int rows_per_sample;
    const float* res;      /* fp32 residual with the addressing of C, or NULL */
    int64_t ldres;
} lnx_gemm_args;

int lnx_gemm_nt(const lnx_gemm_args* args, void* stream);

/* Weight gradient  dW[N,K] += dY[M,N]^T . A[M,K]  and optionally db[N] += colsum(dY).
 * Split over M across workgroups, fp32 atomics into dW/db (caller zeroes them).
 * Replaces autograd's weight/bias gradient of every Linear/patchify conv above. */
typedef struct lnx_wgrad_args {
    int dtype;
    int M, N, K;
    const void* dY; /* [M, N] of T */
    int64_t lddy;
    const void* A;  /* [M, K] of T (PLAIN) or NHWC source (PATCH2) */
    int64_t lda;
    int a_mode, Hin, Win, Cin;
    float* dW;      /* [N, ldw] fp32 */
    int64_t lddw;
    int k_perm_c;   /* >0: K = P*k_perm_c with k = p*C + c stored at column c*P + p
                       (torch conv weight layout [N, C, kh, kw]) */
    float* db;      /* [N] or NULL */
    int splits;     /* 0: choose automatically */
} lnx_wgrad_args;

int lnx_gemm_tn(const lnx_wgrad_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LNX_H */
