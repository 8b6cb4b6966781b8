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
/* Persistent kernels (gemm_nt_v7 / v9, the resident-weight conv-MLP kernels, the depthwise kernels: one workgroup per CU that
 * draws tiles from atomic counters) launch on (compute units - margin) CUs.  Default 0, or LNX_CU_MARGIN from the environment at
 * first use.  With the counters a workgroup that cannot be placed beside a resident collective kernel costs nothing (the others
 * take its tiles); a margin only spares the hardware the queue of unplaceable workgroups.  Replaces nothing in the reference:
 * torch DDP's NCCL kernels and cuBLAS share the SMs the same way (linnaeus/main.py:936-983). */
int lnx_set_cu_margin(int cus);

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
enum { LNX_ACT_NONE = 0, LNX_ACT_GELU = 1, LNX_ACT_RELU = 2, LNX_ACT_GELU_BWD = 3, LNX_ACT_RELU_BWD = 4,
       /* round 3: the derivative is evaluated ONCE, in the forward, where the pre-activation is at hand in fp32 (the polynomial erf
        * and the exponential are the dominant VALU cost of both epilogues; the backward GEMM's becomes one multiply):
        * GELU_D   forward:  C = GELU(v) and c2 = GELU'(v) (instead of v)           -- c2 is required
        * MUL_AUX  backward: v *= aux                                               -- aux = that c2 */
       LNX_ACT_GELU_D = 5, LNX_ACT_MUL_AUX = 6 };
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
    void* c8;              /* lnx_gemm_nt_mxfp8 only, optional: MXFP8 copy of the bf16 output C (exactly lnx_quantize_mxfp8 of C), */
    int64_t ldc8;          /* ldc8 bytes per row, block scales in c8_scales ([N/128][M][4]); needs N % 128 == 0 and the */
    void* c8_scales;       /* bias + GELU + pre-activation form or the GELU' form (the fc1 -> fc2 hand-over of the model's fp8 mode
                              and its mirror image in the backward) */
} lnx_gemm_args;

int lnx_gemm_nt(const lnx_gemm_args* args, void* stream);

/* Several small products in ONE launch (a wave per 32 x 32 output tile, operands straight from L2): what the model's tail needs for its
 * classification heads (mFormerV1.py:536-541) -- the same M = batch product once per head, forward and data gradient.
 *   accumulate == 0: n independent problems, each with its own A / W / C / bias / res / M / N / K (e.g. logits_t = feats . W_t^T + b_t);
 *   accumulate == 1: C = bias + res + sum_j A_j . W_j^T in one accumulator chain (e.g. d feats = sum_t dlogits_t . W_t); M, N, C, ldc,
 *                    bias, res are taken from problem 0, A / lda / W / ldw / K from every problem.
 * bf16 operands, plain addressing, no activation / scales / second output; all problems the same out_f32.  lnx_gemm_nt_group_ok says
 * (host side, no device work) whether a list qualifies; lnx_gemm_nt_group fails loudly on one that does not. */
#define LNX_GEMM_GROUP_MAX 8
int lnx_gemm_nt_group_ok(const lnx_gemm_args* problems, int n, int accumulate);
int lnx_gemm_nt_group(const lnx_gemm_args* problems, int n, int accumulate, void* stream);

/* Which kernel family the NT dispatchers (lnx_gemm_nt, lnx_gemm_nt_fp8, lnx_gemm_nt_mxfp8) chose (host-side bookkeeping, no device work): the parity tests use it to prove
 * that a shape really ran on the kernel it is meant to cover (e.g. the persistent gemm_nt_v7 at the benchmark's M = 50 944).
 * lnx_last_nt_kernel(): family of the most recent NT launch of this process (any stream), 0 before the first.
 * lnx_nt_kernel_launches(kind): launches of that family since the library was loaded. */
enum { LNX_NT_KERNEL_NONE = 0, LNX_NT_KERNEL_V1 = 1 /* 128x128 register-staged, both storage types */, LNX_NT_KERNEL_V2 = 2 /* 256x128 LDS-DMA ring */,
       LNX_NT_KERNEL_SKINNY = 3 /* M <= 256, one wave per tile */, LNX_NT_KERNEL_V4 = 4 /* 256x256 tile */,
       LNX_NT_KERNEL_FP8 = 6 /* lnx_gemm_nt_fp8 / _mxfp8 on the 128x128 tile */, LNX_NT_KERNEL_V7 = 7 /* persistent 256x128, deferred stores */,
       LNX_NT_KERNEL_MX8 = 8 /* lnx_gemm_nt_mxfp8 on the 256x256 tile (gemm_nt_mx8_kernel) */, LNX_NT_KERNEL_V9 = 9 /* persistent 256x256 (round 4) */,
       LNX_NT_KERNEL_EXPERIMENT = 15 /* a kernel of tools/experiments/ (never in the shipped library) */, LNX_NT_KERNEL_KINDS = 16 };
int lnx_last_nt_kernel(void);
int64_t lnx_nt_kernel_launches(int kind);
/* The family lnx_gemm_nt WOULD launch for these arguments (no launch, no device work, no GPU needed; negative on bad arguments): only
 * M / N / K / dtype / out_f32 / act / a_mode / c_mode and WHICH optional operands are non-NULL are looked at, never the memory behind
 * them.  DESIGN.md's dispatch table is checked against this by tests/test_host_logic.py. */
int lnx_nt_dispatch(const lnx_gemm_args* args);

/* fp8 operands (BASELINE config 5's "fp8 MFMA path"; the reference has no fp8 code: this is the MI355X form of its
 * bf16 Linear, mlp.py:46-66 / rope_2d_mhsa.py:432,500).  OCP e4m3fn storage, one dequantisation scale per tensor:
 *   lnx_amax          amax[0] = max(amax[0], max |x|)            (device scalar, caller zeroes it)
 *   lnx_quantize_fp8  y = e4m3(x * 448 / amax[0]),  scale_out[0] = amax[0] / 448   (amax 0 -> scale 1)
 *   lnx_gemm_nt_fp8   C = epilogue( a_scale[0] * w_scale[0] * A8 . W8^T ): `args` as lnx_gemm_nt with dtype = LNX_BF16
 *                     for C / c2 / aux, but A and W are e4m3 bytes (lda / ldw in bytes), K % 128 == 0, N % 16 == 0, M >= 256, plain
 *                     addressing and an epilogue the specialised forms cover (bias, GELU + c2, GELU', fp32 residual).
 * The product runs on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: 2x the bf16 MFMA rate. */
int lnx_amax(const void* x, int x_dtype, int64_t ldx, int rows, int cols, float* amax, void* stream);
int lnx_quantize_fp8(const void* x, int x_dtype, int64_t ldx, int rows, int cols, const float* amax, void* y, int64_t ldy, float* scale_out, void* stream);
int lnx_gemm_nt_fp8(const lnx_gemm_args* args, const float* a_scale, const float* w_scale, void* stream);

/* MXFP8 (OCP microscaling): e4m3 elements, one E8M0 (power-of-two) scale per 32 consecutive elements of a row; the block
 * scales are operands of v_mfma_scale_f32_16x16x128_f8f6f4 itself, so there is no amax pass, no scale state and no
 * dequantisation step -- the CDNA4-native form of the fp8 path.
 *   lnx_quantize_mxfp8  per block: e = smallest exponent with amax * 2^-e <= 448, y = e4m3(x * 2^-e) (round to nearest even),
 *                       scales: bytes [cols/128][rows][4], byte j of (ks, r) = e + 127 of block 4 ks + j of row r
 *                       (cols % 128 == 0; x fp32 or bf16, 16-byte aligned rows; ldy in bytes)
 *   lnx_gemm_nt_mxfp8   C = epilogue( dequant(A8) . dequant(W8)^T ): `args` as lnx_gemm_nt_fp8, a_scales [K/128][M][4],
 *                       w_scales [K/128][N][4] as written by lnx_quantize_mxfp8 */
int lnx_quantize_mxfp8(const void* x, int x_dtype, int64_t ldx, int rows, int cols, void* y, int64_t ldy, void* scales, void* stream);
int lnx_gemm_nt_mxfp8(const lnx_gemm_args* args, const void* a_scales, const void* w_scales, void* stream);

/* Weight gradient  dW[N,K] += dY[M,N]^T . A[M,K]  and optionally db[N] += colsum(dY).
 * Split over M across workgroups.  Partial tiles are added to dW/db with fp32 atomics, or, when the
 * caller passes a workspace (ws), stored there and summed into dW/db by a second kernel in a fixed
 * order (faster: no memory-side atomics; and bit-reproducible).  dW/db accumulate: caller zeroes them.
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
    int k_store;    /* >0: only columns k < k_store are stored (zero-padded K) */
    float* ws;      /* optional split-K workspace (device), NULL = atomics */
    int64_t ws_floats; /* its size; LNX_TN_WS_FLOATS always suffices */
    int defer;      /* round 4.  != 0 (and ws given): the product leaves its split-K partial tiles in ws and the summation into dW / db
                       is postponed to lnx_gemm_tn_flush(), which sums the partial tiles of ALL postponed products of this host thread
                       in one launch (up to 8; a ninth, or one on another stream, flushes first).  Every pending product needs its
                       own ws, untouched until the flush; dW / db are final only after it. */
} lnx_wgrad_args;
#define LNX_TN_WS_FLOATS (256 * (256 * 128 + 256))

int lnx_gemm_tn(const lnx_wgrad_args* args, void* stream);
/* sums the partial tiles of the products postponed with `defer` into their dW / db (no-op when nothing is pending).  `stream` must be the
 * stream those products were launched on (or NULL = that stream): another stream is an error (the reduces would be ordered behind the wrong work). */
int lnx_gemm_tn_flush(void* stream);
/* forgets this host thread's postponed products WITHOUT summing them (returns how many): for a caller whose step failed between a
 * postponed product and its flush -- the descriptors hold raw ws / dW / db pointers that must not outlive those buffers.
 * lnx_plan_backward does this itself on entry and on every error path, lnx_plan_destroy on teardown. */
int lnx_gemm_tn_discard(void);


/* ------------------------------------------------------------------------------------
 * LayerNorm over the last dimension C (biased variance, eps inside the sqrt).
 * In NHWC memory this is also the reference's LayerNormChannelsFirst.
 *   nn.LayerNorm: blocks/convnext.py:59, rope_2d_mhsa.py:546-547, mFormerV1.py:271-273,
 *   294,320-328, res_norm_layer.py:18-19; LayerNormChannelsFirst: blocks/convnext.py:21-43.
 * -----------------------------------------------------------------------------------*/
typedef struct lnx_ln_args {
    int M, C;
    float eps;
    const void* x;      /* rows via x_map, leading dimension ldx */
    int x_dtype;
    int64_t ldx;
    lnx_rowmap x_map;
    const float* w;     /* [C] */
    const float* b;     /* [C] */
    void* y;            /* rows via y_map */
    int y_dtype;
    int64_t ldy;
    lnx_rowmap y_map;
    const void* add;    /* optional [M, C] of x_dtype added to the output (ResNormLayer skip) */
    int64_t ldadd;
    float* mean;        /* [M] or NULL */
    float* rstd;        /* [M] or NULL */
    void* y8;           /* optional second output: MXFP8 copy of the bf16 output (exactly lnx_quantize_mxfp8 of y), compact rows
                           (row m), ldy8 bytes per row, block scales in y8_scales ([C/128][M][4]).  Needs x fp32, y bf16,
                           an identity y_map and C % 128 == 0: the producer side of the model's fp8 mode. */
    int64_t ldy8;
    void* y8_scales;
} lnx_ln_args;
int lnx_layernorm_fwd(const lnx_ln_args* args, void* stream);

typedef struct lnx_ln_bwd_args {
    int M, C;
    const void* dy;     /* rows via dy_map */
    int dy_dtype;
    int64_t lddy;
    lnx_rowmap dy_map;
    const void* x;      /* forward input, rows via x_map */
    int x_dtype;
    int64_t ldx;
    lnx_rowmap x_map;
    const float* w;
    const float* mean;
    const float* rstd;
    const float* gin;   /* optional fp32 gradient to add (rows via x_map, ldgin); may alias dx */
    int64_t ldgin;
    void* dx;           /* rows via x_map */
    int dx_dtype;
    int64_t lddx;
    float* dw;          /* [C] += (fp32 atomics) or NULL */
    float* db;
    int relu_mask;      /* 1: x is a ReLU output feeding this LN; dx *= (x > 0) */
    float* ws;          /* optional scratch for the dw/db column partials (avoids contended atomics) */
    int64_t ws_floats;  /* its capacity in floats; 2048*2*C is the most that is used */
    /* optional second output: dx2[m, :] = dx2_rowscale[m / dx2_rows_per_sample] * dx[m, :] in dx2_dtype, identity rows,
     * leading dimension lddx2.  It is the operand the NEXT branch's GEMMs read (DropPath-scaled gradient in storage
     * type, rope_2d_mhsa.py:630,643 backward), so no separate cast pass re-reads dx.  May alias dy (same dtype/ld). */
    void* dx2;
    int dx2_dtype;
    int64_t lddx2;
    const float* dx2_rowscale;
    int dx2_rows_per_sample;
    /* round 4, optional (fp8 plans with MXFP8 data gradients): the MXFP8 copy of dx2 -- e4m3 elements [M, C] with leading dimension
     * lddx2_8 (bytes = elements) and E8M0 block scales in lnx_quantize_mxfp8's layout for an [M, C] operand -- bit-identical to
     * lnx_quantize_mxfp8 of the bf16 dx2, written by the same pass, so the next branch's data-gradient product needs no quantise pass
     * over its dY.  Needs dx2 in bf16, x in fp32, C % 128 == 0, 4-byte aligned rows. */
    void* dx2_8;
    void* dx2_8_scales;
    int64_t lddx2_8;
    /* round 5.  != 0 (and ws given, dw / db wanted): the column partials stay in ws and their summation into dw / db is postponed to
     * lnx_layernorm_bwd_flush(), which sums the partials of ALL postponed calls of this host thread in one launch (up to 16; a 17th, or
     * one on another stream, flushes first).  Every pending call needs its own ws, untouched until the flush. */
    int defer;
} lnx_ln_bwd_args;
int lnx_layernorm_bwd(const lnx_ln_bwd_args* args, void* stream);
/* sums the postponed column partials into their dw / db (no-op when nothing is pending; `stream` = the postponed calls' stream or NULL) */
int lnx_layernorm_bwd_flush(void* stream);
/* forgets them without summing (error paths; lnx_plan_backward does it on entry and when it fails); returns how many */
int lnx_layernorm_bwd_discard(void);

/* ------------------------------------------------------------------------------------
 * Depthwise 7x7 convolution, padding 3, NHWC (nn.Conv2d(C, C, 7, padding=3, groups=C),
 * blocks/convnext.py:56-58) and its two gradients.  Weights are passed as [49][C] fp32
 * (tap-major; lnx_prep_weights produces this from torch's [C,1,7,7]).
 * -----------------------------------------------------------------------------------*/
typedef struct lnx_dwconv_args {
    int B, H, W, C;       /* C % 32 == 0 */
    const void* x;        /* [B,H,W,C] */
    int x_dtype;
    const float* w49;     /* [49][C] */
    const float* bias;    /* [C] or NULL */
    int flip;             /* 1: correlate with the flipped kernel (data gradient) */
    const float* res;     /* optional fp32 [B,H,W,C] added to the output (may alias y) */
    void* y;              /* [B,H,W,C] */
    int y_dtype;
} lnx_dwconv_args;
int lnx_dwconv7_fwd(const lnx_dwconv_args* args, void* stream);

typedef struct lnx_dwconv_wgrad_args {
    int B, H, W, C;
    const void* x;        /* forward input [B,H,W,C] */
    int x_dtype;
    const void* dy;       /* [B,H,W,C] */
    int dy_dtype;
    float* dw;            /* torch layout [C,1,7,7], += atomics */
    float* db;            /* [C] += or NULL */
} lnx_dwconv_wgrad_args;
int lnx_dwconv7_wgrad(const lnx_dwconv_wgrad_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * Attention with the reference's cos-only "2D RoPE" (SURVEY F1), global over N tokens.
 *   cos table: rope_2d_mhsa.py:56-73,114-155,397-408; q,k scaling :176-218,440-456;
 *   scores/softmax/AV :495-501.  head_dim is 64 for every shipped config.
 * qkv is the raw output of the qkv Linear, [B*N, 3*C] with column = which*C + head*64 + d
 * (:432-437); o is [B*N, C] with column = head*64 + d (:501).  The first E tokens of each
 * sample are extra (CLS/meta) tokens and are not scaled by cos.
 * -----------------------------------------------------------------------------------*/
int lnx_rope_cos_table(const float* freqs /* [2,heads,32] */, int heads, int H, int W, float* cos_out /* [H*W,heads,32] */,
                       float* dsin_out /* optional [2][H*W,heads,32]: -t_x sin(theta), -t_y sin(theta) = d cos(theta) / d freqs[a] (what
                                          lnx_attn_bwd weights its pair gradients with) */,
                       void* stream);
/* The same tables for several blocks in one launch (each RoPE block owns its freqs, rope_2d_mhsa.py:397-408; a plan fills the
 * tables of all its blocks once per forward, off the main stream).  Entries as lnx_rope_cos_table's arguments. */
#define LNX_ROPE_TABLES_MAX 24
typedef struct {
    const float* freqs; /* [2,heads,32] */
    float* cos_out;     /* [H*W,heads,32] */
    float* dsin_out;    /* optional [2][H*W,heads,32] */
    int heads, H, W, pad_;
} lnx_rope_table;
int lnx_rope_cos_tables(const lnx_rope_table* tables, int n, void* stream);
/* floats of lnx_attn_bwd's freqs-gradient workspace (one [2][32] partial per workgroup of its finest tiling) */
int64_t lnx_attn_bwd_ws_floats(int B, int N, int heads);

typedef struct lnx_attn_args {
    int dtype;
    int B, N, E, heads;
    const void* qkv;     /* [B*N, 3*heads*64] */
    const float* cos_tab;/* [(N-E), heads, 32] */
    void* o;             /* [B*N, heads*64] */
    float* lse;          /* [B, heads, N] log-sum-exp of the scaled scores */
    const unsigned char* drop_mask; /* optional (training with MODEL.ATTN_DROP_RATE, rope_2d_mhsa.py:497): keep mask of the attention
                                       probabilities, one byte per (b, head, query, key), [B, heads, N, Np] with Np = N rounded up to
                                       a multiple of 64; P is multiplied by mask * drop_inv_keep AFTER the softmax normalisation.
                                       Runs the 64-row tiled kernels (the dropout-free path keeps its own instantiations). */
    float drop_inv_keep; /* 1 / (1 - ATTN_DROP_RATE) */
} lnx_attn_args;
int lnx_attn_fwd(const lnx_attn_args* args, void* stream);

typedef struct lnx_attn_bwd_args {
    int dtype;
    int B, N, E, heads;
    const void* qkv;
    const float* cos_tab;
    const void* o;
    const float* lse;
    const void* d_o;     /* [B*N, heads*64] */
    void* dqkv;          /* [B*N, 3*heads*64] */
    float* freq_ws;      /* workspace, lnx_attn_bwd_ws_floats(B, N, heads) floats (overwritten): round 3 -- the gradient of the learnable
                            frequencies (autograd of compute_mixed_cis / apply_rotary_emb through the real part only, finding F1) is
                            reduced inside the two backward kernels to one [2][32] partial per workgroup; it used to be a
                            [2][B, N-E, heads, 32] tensor read back by a separate lnx_rope_freqs_bwd */
    float* delta;        /* workspace [B, heads, N] */
    const unsigned char* drop_mask; /* the forward's keep mask (see lnx_attn_args), or NULL */
    float drop_inv_keep;
    const float* dsin_tab; /* [2][(N-E), heads, 32] from lnx_rope_cos_table */
    float* dfreqs;         /* [2, heads, 32] fp32, accumulated into (dfreqs += ...) */
    int defer_freqs;       /* 1: leave the fold of freq_ws into dfreqs to lnx_attn_bwd_flush (one launch for up to LNX_ATTN_DEFER_MAX calls of
                              this thread on this stream; freq_ws and dfreqs must stay untouched and alive until then -- a plan gives every
                              pending call its own freq_ws region and flushes at the end of each backward segment) */
} lnx_attn_bwd_args;
#define LNX_ATTN_DEFER_MAX 16
int lnx_attn_bwd_flush(void* stream); /* folds every postponed call of this thread; refuses a stream other than theirs */
int lnx_attn_bwd_discard(void);       /* forgets them without folding (error paths); returns how many were dropped */
int lnx_attn_bwd(const lnx_attn_bwd_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * Small data-movement / elementwise kernels of the path
 * -----------------------------------------------------------------------------------*/
/* stem im2col: x fp32 NCHW [B,Cin,H,W] -> patches [B*(H/4)*(W/4), ldp] of T with column
 * k = c*16 + kh*4 + kw (torch conv weight order), zero padded to ldp (mFormerV1.py:146) */
int lnx_im2col_stem(const float* x, int B, int Cin, int H, int W, void* patches, int dtype, int ldp, void* stream);

/* The whole stem in one launch (bf16 compute type; csrc/stem.hip): nn.Conv2d(in_chans, dims[0], 4, 4) + bias, output rounded
 * to bf16 as under autocast, then LayerNorm(dims[0], eps, "channels_first") -- mFormerV1.py:145-148.  Writes the fp32
 * normalised rows `y` and, when asked (training plans), what the backward reads: the bf16 patch matrix [M, 64] of
 * lnx_im2col_stem, the bf16 pre-norm tensor [M, Cout] and the row statistics.  lnx_stem_fwd_ok() tells whether the geometry is
 * covered (Cin <= 4, H and W multiples of 4, Cout in {96, 128, 192, 256}); otherwise im2col + lnx_gemm_nt + lnx_layernorm_fwd. */
typedef struct lnx_stem_args {
    const float* x;      /* [B, Cin, H, W] fp32 */
    const void* w;       /* bf16 [Cout, 64], column k = c*16 + kh*4 + kw */
    const float* bias;   /* [Cout] */
    const float* ln_w;   /* [Cout] */
    const float* ln_b;   /* [Cout] */
    void* patches;       /* bf16 [M, 64] or NULL */
    void* pre;           /* bf16 [M, Cout] or NULL */
    float* y;            /* fp32 [M, Cout], M = B * H/4 * W/4 */
    float* mean;         /* [M] or NULL (with rstd) */
    float* rstd;
    int B, Cin, H, W, Cout;
    float eps;
} lnx_stem_args;
int lnx_stem_fwd_ok(int dtype, int Cin, int H, int W, int Cout);
int lnx_stem_fwd(const lnx_stem_args* args, void* stream);

/* out[m, :] = rowscale[m / rows_per_sample] * in[map(m), :]   (fp32 in, T or fp32 out) */
int lnx_scale_cast(const float* in, int64_t ldin, lnx_rowmap in_map, const float* rowscale, int rows_per_sample, void* out,
                   int out_dtype, int64_t ldout, int M, int C, void* stream);

/* LayerScale+DropPath backward of a ConvNeXt block branch (blocks/convnext.py:82-86):
 * dz = rowscale * gamma * g  (T),  dgamma[c] += sum_m rowscale * g * z */
int lnx_layerscale_bwd(const float* g, const void* z, int dtype, const float* gamma, const float* rowscale,
                       int rows_per_sample, void* dz, float* dgamma, int M, int C, void* stream);

/* Round 4.  The LayerScale gradient of a ConvNeXt block WITHOUT the saved pwconv2 output z (blocks/convnext.py:80-83: x = gamma * z,
 * z = act . W2^T + b2).  dgamma[c] = sum_m rs g[m, c] z[m, c] = sum_k W2[c, k] S[c, k] + b2[c] T[c] with S = (rs g)^T act and
 * T = colsum(rs g) -- and the pwconv2 weight / bias gradients are gamma[c] S[c, :] and gamma[c] T[c].  So the weight-gradient product
 * runs on dY = rs g (lnx_convmlp_bwd_args.dz_plain) into ZEROED scratch s [C, K] / t [C] instead of the gradient buffers, and one launch does
 *     dw[c, :] += gamma[c] s[c, :]      db[c] += gamma[c] t[c]      dgamma[c] += sum_k w[c, k] s[c, k] + b[c] t[c]
 * with w, b the fp32 master weight / bias of pwconv2.  No division by gamma (gamma = 0 is an ordinary value), nothing about what the
 * gradient buffers held before.  The forward then no longer writes z (1.4 GB / step at mFormerV1_sm, batch 256) and the backward kernels
 * neither read it nor carry the column sums. */
int lnx_layerscale_apply_wgrad(const float* s, const float* t, int64_t lds, const float* w, const float* b, int64_t ldw, const float* gamma, float* dw, float* db,
                               int64_t lddw, float* dgamma, int C, int K, void* stream);

/* Dropout of the RoPE blocks' Linear outputs (MODEL.DROP_RATE: blocks/mlp.py:61-66 `self.drop`, rope_2d_mhsa.py:503
 * `proj_drop`), with the keep mask drawn by the caller (one byte per element, 1 = keep):
 *   lnx_dropout_mul       x[m, c] = mask[m, c] ? x[m, c] * inv_keep : 0            (in place; T or fp32; C % 8 == 0)
 *   lnx_dropout_residual  out[m, c] = res[m, c] + rowscale[m / rps] * (mask[m, c] ? z[m, c] * inv_keep : 0)   (z of T, fp32 res / out)
 * They are separate passes, not epilogue forms: no shipped configuration trains with dropout, the default path stays as it is. */
int lnx_dropout_mul(void* x, int dtype, const unsigned char* mask, float inv_keep, int M, int C, void* stream);
int lnx_dropout_residual(const void* z, int z_dtype, const unsigned char* mask, float inv_keep, const float* rowscale, int rows_per_sample,
                         const float* res, float* out, int M, int C, void* stream);

/* out[map(m), :] = vec[:]  for m in [0, M)   (CLS token expand, mFormerV1.py:448,487) */
int lnx_fill_rows(const float* vec, float* out, int64_t ldout, lnx_rowmap map, int M, int C, void* stream);
/* out[c] += sum_m in[map(m), c]   (fp32 atomics) */
int lnx_colsum_rows(const float* in, int64_t ldin, lnx_rowmap map, float* out, int M, int C, void* stream);

/* aggregate = Conv1d(2,1,1) over [cls_1, cls_2] (mFormerV1.py:322,515-523):
 * out = w[0]*a + w[1]*b + bias[0] */
int lnx_agg2_fwd(const float* a, const float* b, const float* w2, const float* bias1, float* out, int M, int C, void* stream);
/* da = w0*dout, db = w1*dout, dw[0] += <dout,a>, dw[1] += <dout,b>, dbias += sum(dout) */
int lnx_agg2_bwd(const float* dout, const float* a, const float* b, const float* w2, float* da, float* db_, float* dw2,
                 float* dbias1, int M, int C, void* stream);

/* meta [B, width] fp32 -> T [B, 16] holding columns [off, off+dim) zero padded */
int lnx_pack_meta(const float* meta, int width, int off, int dim, void* out, int dtype, int B, void* stream);

/* ------------------------------------------------------------------------------------
 * Metadata-head chains, one launch per direction (round 5).  Replaces, per metadata component and RoPE stage, the chain
 *     Linear(d, C) -> ReLU -> LayerNorm(C) -> ResNormLayer:  x + LN2(ReLU(W2 . LN1(ReLU(W1 . x))))
 * of /root/reference/linnaeus/models/mFormerV1.py:282-311 and normalization/res_norm_layer.py:23-30 (eps 1e-5 everywhere),
 * and its autograd backward -- until round 4 seven GEMM / LayerNorm launches per head forward and thirteen backward.
 * fp32 storage and fp32 matrix-core arithmetic whatever the plan's dtype (M = batch rows: the cost is launches, not FLOPs).
 * One workgroup takes 16 batch rows through the whole chain; every head passed in one call shares one launch (four per launch).
 * All buffers are the caller's (no allocation, no host synchronisation); rows are 16-byte aligned.
 * ------------------------------------------------------------------------------------ */
typedef struct lnx_meta_head_args {
    int B, C;                 /* batch rows; width of the RoPE stage this head feeds: lnx_meta_heads_supported(C), i.e. C / 128 in {1, 2, 3, 4, 6, 8} */
    int dim, off;             /* this component's input width (1..16) and its first column in `meta` */
    const float* meta;        /* [B, meta_width] */
    int meta_width;
    float eps;                /* LayerNorm eps (the reference: 1e-5) */
    const float* w0;          /* [C, ldw0], columns >= dim zero (the operand arena's copy of `.0.weight`); ldw0 >= 16 */
    int ldw0;
    const float *b0, *ln0_w, *ln0_b;   /* `.0.bias`, `.2.weight`, `.2.bias` */
    const float* w1;          /* [C, ldw1]  `.3.w1.weight` */
    int ldw1;
    const float *b1, *ln1_w, *ln1_b;   /* `.3.w1.bias`, `.3.norm_fn1.*` */
    const float* w2;          /* [C, ldw2]  `.3.w2.weight` */
    int ldw2;
    const float *b2, *ln2_w, *ln2_b;   /* `.3.w2.bias`, `.3.norm_fn2.*` */
    /* activations kept for the backward (also the chain's own scratch: always required) */
    float* t0;                /* [B, 16]  the metadata slice, zero padded */
    float *h0, *x, *h1, *n1, *h2;      /* [B, C] each: ReLU outputs h0 / h1 / h2, LayerNorm outputs x / n1 */
    float *m0, *r0, *m1, *r1, *m2, *r2; /* [B] each: mean / rstd of the three LayerNorms */
    /* output: row b goes to tok[b * tok_row_stride + tok_row_offset .. + C) (the token matrix of the stage: stride N C, offset (1 + m) C) */
    float* tok;
    int64_t tok_row_stride, tok_row_offset;
} lnx_meta_head_args;
/* 1 when the one-launch chain carries this width (heads of one call may have different widths: a launch per run of equal ones) */
int lnx_meta_heads_supported(int C);
int lnx_meta_heads_fwd(const lnx_meta_head_args* heads, int n_heads, void* stream);

typedef struct lnx_meta_head_bwd_args {
    int B, C, dim;
    const float* g;           /* token-matrix gradient: row b at g[b * g_row_stride + g_row_offset .. + C) */
    int64_t g_row_stride, g_row_offset;
    const float* w1t;         /* transposed copies [C_in, ld]: w1t[i][o] = w1[o][i] */
    int ldw1t;
    const float* w2t;
    int ldw2t;
    const float *ln0_w, *ln1_w, *ln2_w;
    const float *t0, *h0, *x, *h1, *n1, *h2, *m0, *r0, *m1, *r1, *m2, *r2;   /* what the forward left */
    float *dp2, *dp1, *dp0;   /* [B, C] scratch each: gradients wrt the three Linear outputs (before the ReLU) */
    float* part;              /* scratch, lnx_meta_heads_bwd_part_floats(B, C) floats: per-row-group column sums of the LayerNorm gradients */
    /* gradients, ACCUMULATED (+=) in a fixed order (no atomics): */
    float* d_w0;              /* [C, dim] (torch layout of `.0.weight`) */
    float *d_b0, *d_ln0_w, *d_ln0_b;
    float* d_w1;              /* [C, C] */
    float *d_b1, *d_ln1_w, *d_ln1_b;
    float* d_w2;              /* [C, C] */
    float *d_b2, *d_ln2_w, *d_ln2_b;
} lnx_meta_head_bwd_args;
int64_t lnx_meta_heads_bwd_part_floats(int B, int C);
/* two launches per (up to four) heads: the data-gradient chain, then every weight / bias / LayerNorm gradient of those heads */
int lnx_meta_heads_bwd(const lnx_meta_head_bwd_args* heads, int n_heads, void* stream);

/* Per-step parameter preparation: cast fp32 master parameters into the T-typed operand
 * arena the GEMMs read (plus transposed copies for the data-gradient GEMMs and the
 * tap-major depthwise weights).  `descs` is a DEVICE array built by the caller. */
enum { LNX_PREP_CAST = 0, LNX_PREP_CONV_PERM = 1, LNX_PREP_DW49 = 2 };
typedef struct lnx_prep_desc {
    const float* src;
    void* dst;        /* [rows, ld] of T (DW49: fp32 [49][C]) */
    void* dst_t;      /* optional transposed copy [cols_out, ld_t] of T, or NULL */
    int rows, cols;   /* source logical shape [rows, cols]; CONV_PERM: cols = C*P */
    int ld, ld_t;
    int P;            /* CONV_PERM: kernel positions per channel (4 for 2x2) */
    int mode;
    int block_start;  /* first workgroup index of this tensor (exclusive prefix sums) */
} lnx_prep_desc;
int lnx_prep_weights(const lnx_prep_desc* descs_dev, int ndesc, int total_blocks, int dtype, void* stream);
/* number of workgroups lnx_prep_weights needs for one tensor */
int lnx_prep_blocks(int rows, int ld, int cols, int ld_t, int has_t);



/* ------------------------------------------------------------------------------------
 * Soft-label cross entropy, forward and logits gradient in one launch ("loss on device").
 * Replaces TaxonomyAwareLabelSmoothingCE.forward (loss/taxonomy_label_smoothing.py:233-405):
 *   loss[b] = class_weight[t_b] * -sum_c soft[t_b, c] * log_softmax(logits[b])[c],  0 where t_b == ignore_index
 * and, with soft == NULL, torch.nn.functional.cross_entropy(..., label_smoothing = smoothing) per sample.
 *   dlogits[b, c] = scale * row_scale[b] * class_weight[t_b] * (softmax(logits[b])[c] * sum_c soft[t_b, c] - soft[t_b, c])
 *   loss_sum     += scale * sum_b row_scale[b] * loss[b]                (atomic; caller zeroes it)
 * -----------------------------------------------------------------------------------*/
typedef struct lnx_softce_args {
    int B, C;
    const float* logits;       /* [B, ld] fp32 */
    int64_t ld;
    const int64_t* target;     /* [B] class indices */
    const float* soft;         /* [C, C] soft-label matrix (rows sum to 1) or NULL = one-hot */
    float smoothing;           /* soft == NULL only: uniform label smoothing epsilon */
    const float* class_weight; /* [C] or NULL */
    int64_t ignore_index;      /* < 0: none */
    const float* row_scale;    /* [B] or NULL: per-sample multiplier of the gradient / of loss_sum */
    float scale;               /* global multiplier (e.g. task_weight / B) */
    float* loss;               /* [B] per-sample losses or NULL */
    float* loss_sum;           /* scalar accumulator or NULL */
    float* dlogits;            /* [B, ldd] or NULL */
    int64_t ldd;
} lnx_softce_args;
int lnx_softce(const lnx_softce_args* args, void* stream);
/* n <= LNX_SOFTCE_MAX_TASKS independent argument sets (the tasks of the multi-task criterion: train.py's per-task loop over
 * `criteria`, loss/hierarchical_loss.py:130-190) in ONE launch */
#define LNX_SOFTCE_MAX_TASKS 8
int lnx_softce_multi(const lnx_softce_args* args, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * GPU-side batch mixing of the collate step: selective Mixup / CutMix and the metadata chunk pick.
 * Replaces linnaeus/aug/gpu/selective_mixup.py:140-230,420-560 and selective_cutmix.py:200-260 (fp32 batches).
 *   mode 0  out[b] = lam x[b] + (1-lam) x[perm[b]]                          (mixup of images / soft targets)
 *   mode 1  out[b] = x[b] with the box [h0,h1) x [w0,w1) taken from x[perm[b]] where valid[b]      (cutmix images)
 *   mode 2  out[b] = valid[b] ? lam x[b] + (1-lam) x[perm[b]] : x[b]                               (cutmix targets)
 * -----------------------------------------------------------------------------------*/
typedef struct lnx_mix_args {
    const float* x;             /* [B, row] */
    const int64_t* perm;        /* [B] partner index (perm[b] == b: unchanged) */
    const unsigned char* valid; /* [B] or NULL (= all valid) */
    float* out;                 /* [B, row], must not alias x */
    int B;
    int64_t row;                /* elements per sample, multiple of 4; mode 1: C*H*W */
    int H, W;                   /* mode 1 only */
    float lam;
    int h0, h1, w0, w1;         /* mode 1 only */
    int mode;
} lnx_mix_args;
int lnx_mix_rows(const lnx_mix_args* args, void* stream);
/* metadata [B, D] with validity mask [B, D] (bytes): per chunk (bounds_dev = device int[2*nchunk] of [start, end)) a chunk
 * with any zero entry counts as absent; both present -> own if pick[b] < 0.5 else the partner's; one present -> it; none ->
 * zeros and an all-false mask */
int lnx_mix_meta(const float* aux, const unsigned char* mask, const int64_t* perm, const float* pick, const int* bounds_dev, int nchunk, int B, int D,
                 float* out_aux, unsigned char* out_mask, void* stream);

/* ------------------------------------------------------------------------------------
 * GPU-side image augmentations of the input pipeline: the tensor operations of the reference's
 * GPUAutoAugmentBatch (linnaeus/aug/gpu/autoaug.py:44-168) and GPURandomErasing (aug/gpu/random_erasing.py:24-94), on fp32
 * [B, C, H, W] batches, each followed by the clamp to [0, 1] the reference applies after every operation; plus the raw-image
 * conversion of h5data/prefetching_h5_dataset.py:214-220.  Which operation / rectangle is chosen stays on the host (the
 * Python classes of linnaeus_amd/aug.py draw exactly as the reference does).
 * -----------------------------------------------------------------------------------*/
enum {
    LNX_AUG_CLAMP = 0,
    LNX_AUG_POSTERIZE = 1,    /* floor(x 255 / p0) p0 / 255, p0 = 2^bits          autoaug.py:117-128 */
    LNX_AUG_SOLARIZE = 2,     /* x < p0 ? x : 1 - x                               :130-131 */
    LNX_AUG_SOLARIZE_ADD = 3, /* x < p1 ? clamp(x + p0) : x                       :133-138 */
    LNX_AUG_INVERT = 4,       /* 1 - x                                            :86 */
    LNX_AUG_BRIGHTNESS = 5,   /* p0 x (what :83-85 names: adjust_brightness)      */
    LNX_AUG_CONTRAST = 6      /* p0 x + (1 - p0) s[image] (adjust_contrast, s = mean grey level: lnx_aug_rowstat kind 1) */
};
/* in place; per_image_scalar: NULL or a device array with one float per image of `per_image` elements (a multiple of 4) */
int lnx_aug_pointwise(float* x, int64_t n, int64_t per_image, int op, float p0, float p1, const float* per_image_scalar, void* stream);
/* in place, three-channel images [B, 3, hw]: c <- clamp(f c + (1 - f) grey)  (Color / Desaturate, autoaug.py:114-115,153-154) */
int lnx_aug_saturation(float* x, int B, int64_t hw, float factor, void* stream);
/* kind 0: out[2r], out[2r+1] = min, max of row r of x [rows, cols]; kind 1: out[r] = mean grey level of RGB image r ([rows, 3, cols]) */
int lnx_aug_rowstat(const float* x, int rows, int64_t cols, int kind, float* out, void* stream);
/* in place: x[r, :] <- clamp((x - min_r) / (max_r - min_r + 1e-6)), minmax from lnx_aug_rowstat kind 0  (AutoContrast / Equalize, :143-151) */
int lnx_aug_rescale(float* x, int64_t rows, int64_t cols, const float* minmax, void* stream);
/* y[., h, w] = x[., round(sy), round(sx)], (sx, sy) = M (w - cx, h - cy) + (cx, cy), M = the host array m6 (2 x 3, output -> source),
 * nearest neighbour, zero outside: torchvision's affine / rotate on tensors with default interpolation and fill (:52-76) */
int lnx_aug_affine(const float* x, float* y, int planes, int H, int W, const float* m6, void* stream);
/* y = clamp(ratio x + (1 - ratio) (x * taps)), taps_dev = device [k, k]; mode 0 reflect padding (GaussianBlurRand, :156-165),
 * mode 1 borders keep x (Sharpness, :140-141) */
int lnx_aug_stencil(const float* x, float* y, int planes, int H, int W, const float* taps_dev, int k, int mode, float ratio, void* stream);
/* rects_dev = device int [n, 5] (image, y0, x0, h, w), values_dev = device float [n, C]: x[image, c, y0:y0+h, x0:x0+w] = value[c] */
int lnx_erase_rects(float* x, int B, int C, int H, int W, const int* rects_dev, const float* values_dev, int n_rects, void* stream);
/* raw image batch uint8 [B, H, W, C] -> fp32 [B, C, H, W] / 255 on the device (a quarter of the PCIe bytes of a float batch) */
int lnx_u8hwc_to_f32chw(const unsigned char* src, float* dst, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimizer + step glue: multi-tensor AdamW with the global-norm clip folded in.
 * Replaces torch.optim.AdamW.step as configured by linnaeus/optimizers/build.py:307-686 (per-group lr / weight
 * decay) and the gradient-norm passes of train.py:282-308 (clip_grad_norm_ semantics:
 * coef = min(1, max_norm / (||g||_2 + 1e-6))).  `descs` is a DEVICE array, one entry per parameter tensor.
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/bias_c1 * m / (sqrt(v)/sqrt(bias_c2) + eps)
 * -----------------------------------------------------------------------------------*/
#define LNX_ADAMW_MAX_GROUPS 16
typedef struct lnx_adamw_desc {
    float* p;         /* parameter (fp32 master) */
    float* m;         /* exp_avg */
    float* v;         /* exp_avg_sq */
    const float* g;   /* gradient */
    int64_t n;        /* elements */
    int group;        /* index into lnx_adamw_hyper */
    int block_start;  /* first workgroup of this tensor: exclusive prefix sum of lnx_adamw_blocks(n) */
} lnx_adamw_desc;
typedef struct lnx_adamw_hyper {
    int ngroups;
    float lr[LNX_ADAMW_MAX_GROUPS], beta1[LNX_ADAMW_MAX_GROUPS], beta2[LNX_ADAMW_MAX_GROUPS], eps[LNX_ADAMW_MAX_GROUPS],
        weight_decay[LNX_ADAMW_MAX_GROUPS];
    float bias_c1[LNX_ADAMW_MAX_GROUPS], bias_c2[LNX_ADAMW_MAX_GROUPS]; /* 1 - beta^step of the step being taken */
    float omb1[LNX_ADAMW_MAX_GROUPS], omb2[LNX_ADAMW_MAX_GROUPS];       /* 1 - beta, rounded from double (1 - 0.999f loses 1e-5) */
} lnx_adamw_hyper;
int lnx_adamw_blocks(int64_t numel);
/* out[0] = sum over all tensors of g^2.  ws: total_blocks floats (one partial per workgroup, folded in a fixed order by a second
 * launch): the same gradients give the same bits on every data-parallel rank, so the clip coefficient -- and with it the
 * replicas' parameters -- cannot drift apart (clip_grad_norm_ of train.py:282-308 is deterministic in the same sense). */
int lnx_grad_sumsq(const lnx_adamw_desc* descs_dev, int ndesc, int total_blocks, float* out, float* ws, void* stream);
/* sumsq == NULL or max_norm <= 0: no clipping */
int lnx_adamw_step(const lnx_adamw_desc* descs_dev, int ndesc, int total_blocks, const lnx_adamw_hyper* hyper, const float* sumsq, float max_norm,
                   void* stream);

/* ------------------------------------------------------------------------------------
 * Fused ConvNeXt MLP branch (bf16 storage, C in {32,64,96,128,192}):
 *   out = x + rowscale * gamma * (GELU(ln . W1^T + b1) . W2^T + b2)
 * = pwconv1 -> GELU -> pwconv2 -> LayerScale -> DropPath -> residual (blocks/convnext.py:79-86)
 * with the 4C hidden activation kept on chip; the backward recomputes it per tile.
 * -----------------------------------------------------------------------------------*/
int lnx_convmlp_supported(int dtype, int C);
typedef struct lnx_convmlp_args {
    int dtype, M, C;
    const void* ln;        /* [M, C] bf16 (block LayerNorm output) */
    const void* w1;        /* [4C, C] bf16 pwconv1.weight */
    const float* b1;       /* [4C] */
    const void* w2;        /* [C, 4C] bf16 pwconv2.weight */
    const float* b2;       /* [C] */
    const float* gamma;    /* [C] */
    const float* rowscale; /* per-sample DropPath multiplier or NULL */
    int rows_per_sample;
    const float* x;        /* [M, C] fp32 residual input */
    float* out;            /* [M, C] fp32 */
    void* z;               /* optional [M, C] bf16: pwconv2 output before gamma (for the gamma gradient) */
    /* Fused block LayerNorm (blocks/convnext.py:77,84 `x = self.norm(x)`, eps 1e-6; round 3): with `y` set the kernel reads the
     * LayerNorm INPUT (the depthwise conv output) and normalises it in registers -- `ln` must then be NULL -- and writes what the
     * backward needs: the normalised rows (operand of the pwconv1 weight gradient) and the row statistics. */
    const void* y;         /* [M, C] bf16, or NULL: `ln` is given */
    const float* ln_w;     /* [C] */
    const float* ln_b;     /* [C] */
    float ln_eps;
    void* ln_out;          /* optional out [M, C] bf16 */
    float* mean;           /* optional out [M] (with rstd) */
    float* rstd;
} lnx_convmlp_args;
int lnx_convmlp_fwd(const lnx_convmlp_args* args, void* stream);

typedef struct lnx_convmlp_bwd_args {
    int dtype, M, C;
    const float* g;        /* [M, C] fp32 gradient of the block output */
    const void* ln;        /* [M, C] bf16 */
    const void* z;         /* [M, C] bf16 saved by the forward, or NULL (round 4): dgamma is then NOT produced here -- see
                              lnx_layerscale_apply_wgrad */
    const void* w1;        /* [4C, C] bf16 */
    const float* b1;
    const void* w2t;       /* [4C, C] bf16 = pwconv2.weight^T */
    const void* w1t;       /* [C, 4C] bf16 = pwconv1.weight^T */
    const float* gamma;
    const float* rowscale;
    int rows_per_sample;
    void* act;             /* out [M, 4C] bf16  GELU(h)   (operand of the pwconv2 weight gradient), or NULL with dh NULL: */
    void* dh;              /* out [M, 4C] bf16  dL/dh     (operand of the pwconv1 weight gradient)   not materialised  */
    void* dz;              /* out [M, C]  bf16  rowscale*gamma*g */
    void* dln;             /* out [M, C]  bf16  gradient wrt the LayerNorm output */
    float* dgamma;         /* [C] += */
    /* Fused LayerNorm backward (round 3): with `y` set, `dln` receives the gradient wrt the LayerNorm INPUT y instead (what
     * lnx_layernorm_bwd would make of dln, y, mean, rstd), and d_ln_w / d_ln_b += the LayerNorm weight / bias gradient. */
    const void* y;         /* [M, C] bf16, or NULL */
    const float* ln_w;     /* [C] */
    const float* mean;     /* [M] saved by the forward */
    const float* rstd;
    float* d_ln_w;         /* [C] += */
    float* d_ln_b;         /* [C] += */
    float* ws;             /* scratch for the per-workgroup column sums: lnx_convmlp_bwd_ws_floats(C, M) floats (2 C per workgroup */
    int64_t ws_floats;     /* of the launch this library would make for (C, M)); too small = error                              */
    int dz_plain;          /* round 4.  != 0: the `dz` written to memory is rowscale * g, WITHOUT the LayerScale factor gamma (inside the
                              kernel the data gradient keeps it): the dY operand of a pwconv2 weight-gradient product whose result
                              lnx_layerscale_apply_wgrad multiplies by gamma afterwards (and reads the LayerScale gradient from) */
    int ln_defer;          /* round 5.  != 0 (fused LayerNorm form): the fold of the workgroups' column sums into d_ln_w / d_ln_b is postponed to
                              lnx_layernorm_bwd_flush() together with the postponed LayerNorm backward calls; `ws` must stay untouched until then */
} lnx_convmlp_bwd_args;
int lnx_convmlp_bwd(const lnx_convmlp_bwd_args* args, void* stream);
/* floats of `ws` the fused-LayerNorm backward needs for (C, M), from the same grid choice the launcher makes (0: unsupported C) */
int64_t lnx_convmlp_bwd_ws_floats(int C, int M);

/* (round 2 had lnx_convmlp_wgrad here: both weight gradients from ln / dz with the hidden recomputed on chip, so that act / dH
 * never reached HBM.  Correct, but slower end to end on MI355X -- 302 + 207 us per block at C = 96 against 317 us for the two
 * weight-gradient GEMMs it replaced, because a single launch cannot hold the 288 KB of dW1 + dW2 accumulators and each of the two
 * launches re-evaluates the GELU -- and removed in round 3; DESIGN.md records the numbers.) */

/* ------------------------------------------------------------------------------------
 * Whole-model plan: mFormerV1 forward and backward as one native call each.
 * Replaces mFormerV1.forward_features/forward (models/mFormerV1.py:407-541) and its
 * autograd backward.  The plan owns no memory: the caller passes one workspace of
 * lnx_plan_workspace_bytes() bytes (activations saved for backward, the T-typed operand
 * arena, scratch) and binds parameter / gradient pointers in the plan's parameter order
 * (= the reference's state_dict order, see lnx_plan_param_name).
 * -----------------------------------------------------------------------------------*/
#define LNX_MAX_META 8
#define LNX_MAX_TASKS 16
typedef struct lnx_mformer_cfg {
    int dtype;                 /* LNX_F32 (strict parity) or LNX_BF16 */
    int batch, img_h, img_w, in_chans;
    int dims[4];               /* CONVNEXT_STAGES.DIMS (dims[2:] are the RoPE dims) */
    int conv_depths[2];        /* CONVNEXT_STAGES.DEPTHS[0:2] */
    int rope_depths[2];
    int rope_heads[2];
    int mlp_hidden[2];         /* int(dim * MLP_RATIO) */
    int n_meta;                /* enabled metadata components, 0 = metadata inactive */
    int meta_dims[LNX_MAX_META];
    int only_last_cls;
    int n_tasks;
    int task_classes[LNX_MAX_TASKS];
    int inference;             /* 1: forward-only plan (model.eval() under no_grad, validation.py:199): no backward scratch,
                                  blocks share their activation buffers; lnx_plan_backward fails on it */
    int recompute;             /* 1: activation recompute (gradient checkpointing, blocks/convnext.py:89-100,
                                  rope_2d_mhsa.py:617-641): only each block's INPUT is kept; the blocks of a stage share one
                                  set of activation buffers and lnx_plan_backward re-runs a block's forward right before its
                                  backward.  Same results bit for bit (the recompute is deterministic and reuses the DropPath
                                  draw of the forward); workspace no longer grows with the depth. */
    int fp8;                   /* 1 (with dtype = LNX_BF16): the forward products of the RoPE blocks' qkv / fc1 / fc2 Linear layers run
                                  on the block-scaled fp8 matrix cores (MXFP8: lnx_quantize_mxfp8 + lnx_gemm_nt_mxfp8) -- BASELINE
                                  config 5's "fp8 MFMA path".  Weights are re-quantised from the fp32 masters every forward,
                                  activations as they are produced; everything saved for the backward, and the backward, stay
                                  bf16 (environment LNX_FP8_DGRAD=1: the proj / fc2 / fc1 data-gradient products in MXFP8 as
                                  well, gradients quantised per 32-element block, dY as the MXFP8 copy the LayerNorm backward
                                  writes beside its bf16 output -- measured: +1 % speed at xl for a quarter more gradient
                                  error, so off by default).  Needs RoPE dims and MLP widths that are multiples
                                  of 128. */
} lnx_mformer_cfg;

typedef struct lnx_plan lnx_plan;
int lnx_plan_create(const lnx_mformer_cfg* cfg, lnx_plan** out);
void lnx_plan_destroy(lnx_plan* p);
int64_t lnx_plan_workspace_bytes(const lnx_plan* p);
int lnx_plan_num_params(const lnx_plan* p);
/* "stem.0.weight", "stages.2.0.attn.freqs", ...; heads are "head.<task index>.weight|bias" */
const char* lnx_plan_param_name(const lnx_plan* p, int i);
int64_t lnx_plan_param_numel(const lnx_plan* p, int i);
int lnx_plan_num_drop_calls(const lnx_plan* p);
/* logits are written as one fp32 buffer: task t occupies [batch, ld_t] at element offset off_t */
int64_t lnx_plan_logits_numel(const lnx_plan* p);
int64_t lnx_plan_logits_offset(const lnx_plan* p, int task);
int lnx_plan_logits_ld(const lnx_plan* p, int task);
/* Bind device pointers: params[i] / grads[i] fp32 in plan order (grads may be NULL for an
 * inference-only plan), workspace of lnx_plan_workspace_bytes().  Synchronises the device once. */
int lnx_plan_bind(lnx_plan* p, const float* const* params, float* const* grads, void* workspace);
/* x [B,Cin,H,W] fp32 NCHW, meta [B, sum(meta_dims)] fp32 or NULL, drop_scales [n_drop_calls, B]
 * fp32 per-sample DropPath multipliers (NULL = eval / no DropPath), drop_mask[i] != 0 selects
 * which calls use their row.  Outputs: feats [B, dims[3]] fp32, logits (layout above). */
int lnx_plan_forward(lnx_plan* p, const float* x, const float* meta, const float* drop_scales, const unsigned char* drop_mask,
                     float* feats, float* logits, void* stream);
/* Training-time dropout of the RoPE blocks (MODEL.DROP_RATE): keep masks for the next forward and its backward, one byte per
 * element, per RoPE block (stage 3 blocks first) [proj output M x C][MLP hidden M x hidden][fc2 output M x C] -- the caller
 * draws them (Bernoulli(1 - drop_rate)) and keeps the buffer alive until the backward has run.  masks = NULL or
 * drop_rate = 0 switches dropout off.  Not available on fp8 or inference plans. */
int64_t lnx_plan_dropout_bytes(const lnx_plan* p);
int lnx_plan_set_dropout(lnx_plan* p, const unsigned char* masks, float drop_rate);
/* The same for MODEL.ATTN_DROP_RATE (dropout on the attention probabilities, rope_2d_mhsa.py:497): per RoPE block a keep
 * mask [B, heads, N, Np] (Np = N rounded up to a multiple of 64), see lnx_attn_args.drop_mask. */
int64_t lnx_plan_attn_dropout_bytes(const lnx_plan* p);
int lnx_plan_set_attn_dropout(lnx_plan* p, const unsigned char* masks, float drop_rate);
/* Backward of the last forward.  dlogits has the logits layout; dfeats [B, dims[3]] is an
 * optional extra gradient on feats (NULL).  Parameter gradients are ACCUMULATED into the bound
 * grads.  segment: -1 = everything, or 0..3 = {tail + RoPE stage 4, RoPE stage 3, ConvNeXt
 * stage 2, ConvNeXt stage 1 + stem}, to be called in that order (lets the caller start the
 * gradient all-reduce of a finished segment while the next one runs). */
int lnx_plan_backward(lnx_plan* p, const float* dlogits, const float* dfeats, int segment, void* stream);
/* Live per-kernel-class timing with HIP events on the launch stream (used by bench.py for the
 * roofline line; adds event records around the timed launches, so never leave it on in a timed
 * region).  Classes: 0 gemm_nt, 1 gemm_tn, 2 attention fwd, 3 attention bwd (both kernels),
 * 4 depthwise conv fwd / data-grad, 5 depthwise conv weight-grad, 6 fused conv-MLP forward,
 * 7 fused conv-MLP backward (data side), 8 fused conv-MLP weight gradients.  work = FLOPs for 0-3 and 6-8,
 * algorithmic HBM bytes for 4-5. */
#define LNX_PROFILE_CLASSES 8
int lnx_plan_profile_begin(lnx_plan* p);
int lnx_plan_profile_end(lnx_plan* p, double* ms, double* work, int* launches);
/* Round 4.  lnx_plan_profile_end_ex: as lnx_plan_profile_end with LNX_PROFILE_CLASSES_EX entries per array and, per class, the
 * ALGORITHMIC HBM bytes of its launches (every operand read once, every output written once; GEMM classes 0 / 1 only) -- what
 * bench.py's roofline object prices the achieved TB/s and the FLOP/byte of the dominant class with.
 * lnx_plan_profile_begin_spans: block-level timing instead of per-launch timing -- ONE event pair around each whole block on
 * the main stream, forward and backward (incl. a recompute plan's re-forward): class 8 = RoPE2DMHSABlock (rope_2d_mhsa.py:584-645:
 * both LayerNorms, qkv, attention, proj, Mlp, their weight gradients and reduces), work = the block's FLOPs (forward 1x +
 * backward 2x of its GEMM and attention products: BASELINE.md's 15.94 GFLOP/img for mFormerV1_sm); class 9 = ConvNeXtBlock
 * (convnext.py:73-87).  No events sit between the kernels of a block, so the span is what the block costs in the timed step. */
#define LNX_PROFILE_CLASSES_EX 10
int lnx_plan_profile_begin_spans(lnx_plan* p);
int lnx_plan_profile_end_ex(lnx_plan* p, double* ms, double* work, double* bytes, int* launches);
/* Round 4: the backward's weight-gradient stream.  By default a plan runs the RoPE blocks' weight-gradient products (+ their batched
 * split-K reduce) and the fused ConvNeXt blocks' pointwise weight gradients (+ the LayerScale step) on a HIP stream of its own, forked
 * behind the kernel that wrote each product's dY and joined before that buffer's next writer and at the end of every block -- the
 * same kernels, the same sums in the same order (bit-equal gradients), ramps and tails of one chain under the body of the other.
 * lnx_plan_set_wgrad_stream(p, 0) puts everything back on the launch stream (what bench.py's per-kernel timing pass does, so that a
 * kernel's duration is its own); returns the previous setting (0 / 1), negative on error.  LNX_WGRAD_STREAM=0 never creates the stream. */
int lnx_plan_set_wgrad_stream(lnx_plan* p, int on);
/* Round 5: which stream the metadata heads (forward chain, backward chain + weight gradients) run on beside the launch stream:
 * 0 = the launch stream itself, 1 = a side stream of their own (default; LNX_NO_SIDE_STREAM never creates it), 2 = the weight-gradient
 * stream (one HIP stream fewer: a data-parallel step then uses launch + weight-gradient streams + the collective library's own --
 * within the four hardware queues of the device; only with the one-launch chain, otherwise as 1).  Forks and joins are events either
 * way: same values, same summation orders.  Returns the previous mode, negative on error.  LNX_META_STREAM=0/1/2 sets the default. */
int lnx_plan_set_meta_stream(lnx_plan* p, int mode);
/* indices of the parameters whose gradient is final after `segment`; returns their count.  The metadata heads' backward
 * runs on the plan's side stream and is joined one segment after the one that forks it, so the stage-4 heads report
 * segment 1 and the stage-3 heads segment 2 (a caller that stops early must run the following segment, or -1, to join). */
int lnx_plan_segment_params(const lnx_plan* p, int segment, int* idx_out, int max_out);

#ifdef __cplusplus
}
#endif
#endif /* LNX_H */
