/*
 * npm_hip.h -- C ABI of the MI355X (gfx950) layer forward/backward hot path.
 *
 * This is the drop-in boundary.  The reference
 * (levendlee/np-modeling) is pure NumPy: it has no FFI, so every entry point below
 * replaces a NumPy call site instead of a native symbol; the site is cited as
 * reference file:line.  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns 0 on success or a
 *     non-zero hipError_t / NPM_E_* code, and npm_last_error() describes the failure.
 *   - One process drives one GPU: npm_init(device) binds the process to a device and
 *     creates the compute stream all launches go to.  Launches are asynchronous;
 *     npm_sync() / npm_d2h() are the synchronisation points.
 *   - ONE host thread per process: the library keeps process-global state (the device, its one compute
 *     stream, the caching pool, the math mode and tuning knobs, npm_last_error / npm_last_math /
 *     npm_last_attn_kernel) without locks.  Calls from several threads must be serialised by the caller;
 *     parallelism across GPUs is one process per GPU (np_modeling_amd/launch.py, include/npm_comm.h).
 *   - All tensors are fp32, row-major; "ld" is the row pitch in elements.
 *   - Device pointers come from npm_malloc (a stream-ordered caching pool).
 */
#ifndef NPM_HIP_H
#define NPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPM_ABI_VERSION 2      /* 2: npm_mha_core gained tile_summary .. summary_all_offset, npm_comm_exchange_stats last_allreduce_ms / dropped */

enum {
    NPM_OK = 0,
    NPM_E_NOT_INITIALIZED = 10001,
    NPM_E_BAD_ARGUMENT = 10002,
    NPM_E_UNSUPPORTED = 10003,
    NPM_E_NO_DEVICE = 10004
};

/* ---- runtime ------------------------------------------------------------ */
int npm_abi_version(void);
const char *npm_last_error(void);
int npm_device_count(int *count);
int npm_init(int device);                 /* idempotent for the same device */
int npm_shutdown(void);
int npm_device_name(char *buf, int len);
void *npm_stream(void);                   /* the hipStream_t launches go to */
int npm_sync(void);                       /* hipStreamSynchronize(compute stream) */

/* ---- memory: caching pool over hipMalloc --------------------------------- */
int npm_malloc(void **ptr, size_t bytes);
int npm_free(void *ptr);                  /* returns the block to the pool (stream-ordered reuse) */
int npm_pool_stats(size_t *bytes_in_use, size_t *bytes_reserved);
int npm_pool_trim(void);                  /* hipFree every cached block */
int npm_h2d(void *dst, const void *src, size_t bytes);   /* ordered after prior launches; returns when done */
int npm_d2h(void *dst, const void *src, size_t bytes);   /* ditto */
int npm_d2d(void *dst, const void *src, size_t bytes);   /* async on the compute stream */
int npm_fill_f32(float *dst, float value, size_t n);

/* ---- events (HIP events on the compute stream; used by bench.py) ---------- */
int npm_event_create(void **event);
int npm_event_destroy(void *event);
int npm_event_record(void *event);
int npm_event_sync(void *event);
int npm_event_elapsed_ms(void *start, void *stop, float *ms);

/* ---- GEMM: C = epilogue(alpha * op(A) op(B)) ------------------------------
 * Replaces np.matmul in layers/mlp.py:23,35,36 and every np.einsum contraction in
 * layers/attentions.py:88-117,129-188 (each is a flat or batched GEMM view).
 * fp32 operands on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate.
 *
 *   trans_a == 0 : A is [M,K] (pitch lda)      trans_a == 1 : A is stored [K,M]
 *   trans_b == 0 : B is [K,N] (pitch ldb)      trans_b == 1 : B is stored [N,K]
 * Batches are indexed z = z0 * batch1 + z1 with operand offset z0*stride0 + z1*stride1
 * (lets [B,S,H,D] head slices be addressed without a transpose).
 */
enum {
    NPM_EPI_BIAS = 1,        /* + bias[n]                                  (mlp.py:24) */
    NPM_EPI_RESIDUAL = 2,    /* + residual[m,n] (may alias C: accumulate)  (transformer.py:39,53) */
    NPM_EPI_RELU_SAVE = 4,   /* aux[m,n] = v; C = max(v,0)                 (activations.py:14-15) */
    NPM_EPI_RELU_MASK = 8,   /* C = aux[m,n] >= 0 ? v : 0                  (activations.py:19) */
    NPM_EPI_RELU = 16,       /* C = max(v,0), pre-activation not kept (inference) */
    NPM_EPI_SOFTMAX_BWD = 32,/* C = alpha * aux[m,n] * (acc - rowvec[m]): softmax backward with the row term
                                sum_j dP_ij P_ij = dctx_i . ctx_i precomputed (activations.py:32-45, attentions.py:150-155) */
    NPM_EPI_ROWDOT = 128     /* C = acc (stored as is) AND rowdot[(n / 128) * m_total + i] += rowdot_scale * sum over the 128 columns
                                j of column block n / 128 of C[i, j] * aux[i, j]: the row term delta_i = dctx_i . ctx_i of the attention
                                backward (attentions.py:150-155, the Jacobian-vector product of Softmax.backward) per head of size
                                128, taken where dctx = dy wo is produced (attentions.py:136) instead of by a pass over dctx and
                                ctx.  `rowdot` ([n / 128, m]) must be ZERO on entry (two partial sums per element are added with
                                float atomics: commutative, so bitwise reproducible).  Plain product only: no other epilogue flag,
                                no batch, no split-K, n % 128 == 0, the LDS-DMA kernel's alignment rules; else NPM_E_UNSUPPORTED. */
};

typedef struct npm_gemm {
    int32_t trans_a, trans_b;
    int32_t m, n, k;
    int32_t batch0, batch1;              /* >= 1 each */
    const float *a; int64_t lda, stride_a0, stride_a1;
    const float *b; int64_t ldb, stride_b0, stride_b1;
    float *c;       int64_t ldc, stride_c0, stride_c1;
    float alpha;
    int32_t epilogue;                    /* NPM_EPI_* bit set */
    const float *bias;                   /* [n] */
    const float *residual; int64_t ldr;  /* same batch strides as C */
    float *aux; int64_t ldaux;           /* same batch strides as C */
    int32_t split_k;                     /* 0 = choose automatically, 1 = never split */
    const float *rowvec;                 /* [batch, m] per-row term of NPM_EPI_SOFTMAX_BWD */
    float *colsum;                       /* optional [batch1, n]: colsum[z1, j] = sum over z0 and rows of the stored C
                                            (the bias gradient np.sum(dy, axis=0), mlp.py:34 / attentions.py:190-197,
                                            taken in the producing GEMM's epilogue; fixed summation order) */
    float *bsum;                         /* optional [n]: bsum[j] = sum over k of B[k, j] -- the column sums of the
                                            second operand of a weight-gradient product x^T dy, i.e. the bias gradient
                                            that goes with it (mlp.py:34-35, attentions.py:129-135,190-197), taken from
                                            the B tiles the GEMM stages anyway.  Needs trans_b = 0, batch0 = batch1 = 1. */
    float *asum;                         /* optional [m]: asum[i] = sum over k of A[k, i] for a transposed A (trans_a = 1,
                                            stored [k, m]): the same for products written dproj^T x, whose bias gradient
                                            sums the FIRST operand (attentions.py:167-197).  Not together with bsum. */
    float *rowdot; float rowdot_scale;   /* NPM_EPI_ROWDOT (ABI version 2) */
} npm_gemm;

int npm_sgemm(const npm_gemm *g);

/* Tuning knobs for A/B experiments in one process (tools/gemm_bench.py --tune, NPM_TUNE=knob=value,...).
 * Defaults are the shipped configuration: LDS-DMA pipeline (2), tile-row groups of 8, buffer epilogue on,
 * convolution DMA on.  NPM_TUNE_GEMM_ABLATE is a timing-only diagnostic: it skips work and breaks results. */
enum {
    NPM_TUNE_GEMM_PIPELINE = 0,      /* 0 register-staged 2 barriers, 1 register-staged double buffer, 2 LDS-DMA */
    NPM_TUNE_GEMM_GROUP_M = 2,
    NPM_TUNE_GEMM_BUF_EPILOGUE = 3,
    NPM_TUNE_CONV_DMA = 4,
    NPM_TUNE_GEMM_WIDE_TILE = 5,     /* 128 x 256 block tile (8 waves) where n % 256 == 0: 0 never (default), 1 always, 2 NN / NT, 3 NT only */
    NPM_TUNE_LN_BWD_BLOCKS = 6,      /* blocks per CU of the LayerNorm backward grid (default 4) */
    NPM_TUNE_EW_GRID_CAP = 7,        /* max blocks of the grid-stride elementwise kernels (default 2^20) */
    NPM_TUNE_CONV_WGRAD_BLOCKS = 8,  /* grad_w split-K blocks per CU: 0 (default) best of 3 and 4, 3 / 4 pinned, 6 / 9 / 12 several generations of shorter K ranges (measured at C3: 14.44 -> 14.6-15.0 ms), -1 unbalanced ceil(3 CUs / tiles) */
    NPM_TUNE_GEMM_WAVE_PRIO = 9,     /* s_setprio 3 in the GEMM / conv block prologue (bit 0) and epilogue (bit 1) */
    NPM_TUNE_GEMM_MATH = 10,         /* same as npm_set_math */
    NPM_TUNE_ATTN_STAGGER = 11,      /* attention forward: s_sleep(127) units one of the two blocks of a CU waits at its start (default 1) */
    NPM_TUNE_CONV_WGRAD_FUSED = 13,  /* npm_conv2d_bwd_w_relu: 1 (default) ReLU backward inside the grad_w kernel, tile height picked (k*k*C0 of exactly three 192-row tiles: one block of twelve waves sharing the masked dy tile); 2 / 3 the same with 128- / 192-row tiles in four-wave blocks; 0 two passes */
    NPM_TUNE_ATTN_BWD16 = 14,        /* attention backward: 2 (default) mha_bwd8_kernel (8 waves on the 16x16x4 MFMA, one barrier per tile, every head size, both score modes, tile skipping) except head size 128 with saved scores and no tile summary, which runs mha_bwd16_kernel; 3 mha_bwd8_kernel always; 1 round 3's choice (mha_bwd16_kernel for head size 128 with saved scores, the 4-wave 32x32x2 kernel otherwise); 0 the 4-wave kernel always */
    NPM_TUNE_KSYNC = 15,             /* K tiles between the soft rendezvous of the co-resident split-K blocks of the fused Conv2D filter gradient: a power of two, default 128; 0 off */
    NPM_TUNE_CONV_KORDER = 16,       /* Conv2D forward / grad_x K loop: 1 (default) the k k taps of one 16-channel chunk back to back (the lines a tap fetched are still in L2 when its neighbour wants them: grad_x of C3 reads 13.8 instead of 82 GB past the L2s, +4 %), 0 taps outermost (kk = tap C + c) */
    NPM_TUNE_ATTN_FWD8 = 17,         /* attention forward: 2 mha_fwd8_kernel (8 waves per block on the 16x16x4 MFMA, four waves per SIMD) for every head size; 1 below head size 128 only; 0 the 4-wave 32x32x2 mha_fwd_kernel always */
    NPM_TUNE_GEMM_SPLIT_GENS = 18,   /* split-K of tall-K products (weight gradients): 1 (default) for A-heavy products that also sum A's columns, a K range longer than 768 K tiles is cut further when that makes whole generations of resident blocks (3 x 4 per CU: the packed q/k/v weight gradient 6.00 -> 5.71 ms); 0 one generation of three blocks per CU always (round 3) */
    NPM_TUNE_STREAM_NT = 12,         /* 1 (default): the HBM-bound kernels move tensors of >= 32 MB with the nontemporal cache hint; 0: default policy */
    NPM_TUNE_LN_NT_SPLIT = 19,       /* LayerNorm at d in (512, 1024]: backward mode + 4 * forward mode; a mode: 0 nontemporal hint on loads and stores, 1 on the loads only, 2 on the stores only.  Default 5: loads only in both (dx and z are read at once by the GEMMs behind them; measured inside the encoder step, profiles/r05_ln_nt_split.log) */
    NPM_TUNE_GEMM_ABLATE = 99
};
int npm_set_tuning(int knob, int value);

/* Arithmetic of the matrix products (GEMM and convolution kernels).  Inputs, outputs and accumulators are fp32 in
 * every mode; what changes is the instruction that forms the products:
 *   NPM_MATH_F32          v_mfma_f32_32x32x2_f32: exact fp32 products, bit-equal to a k-ordered fmaf chain (default).
 *   NPM_MATH_BF16X3       each operand is split in registers into three bf16 parts (hi rounded, mid, lo: 24 mantissa
 *                         bits, the parts add up to the fp32 value) and the product is formed as the six largest
 *                         cross terms on v_mfma_f32_32x32x16_bf16, the five small terms in accumulators of their own.
 *                         Error against fp64 is at or below the f32 mode's (tools/math_bias.py);
 *                         results are NOT bit-equal to the f32 mode.  Non-finite operands: an infinity (or a
 *                         value that rounds to one in bf16, |a| > 3.389e38) splits into inf + (inf - inf) and gives
 *                         NaN where the f32 mode gives inf; operands below 2^-110 lose their low parts to underflow.
 *   NPM_MATH_BF16X3_FAST  the same six terms into one accumulator: fewer registers, faster; the matrix pipe cuts
 *                         small addends against a large accumulator, which leaves a bias of about -0.5 ulp per
 *                         4096 accumulated terms (visible in column checksums, not per element).
 *   NPM_MATH_F16X2        two-way fp16 split with ROW SCALING on v_mfma_f32_32x32x16_f16, three MFMAs per product: every
 *                         row of op(A) and column of op(B) is scaled by a power of two that brings its largest magnitude
 *                         along K to [2^13, 2^14), split into hi + lo (11 + 11 bits), and the product is
 *                         (hi hi + hi lo + lo hi) / (s_a s_b).  The error is fp32-class ROW-NORMWISE (rms 1.4e-7 of the
 *                         output row's largest element at K = 4096, below a k-ordered fp32 fma chain's 3.8e-7) rather
 *                         than elementwise: an operand element more than 2^17 below its row's maximum keeps fewer than
 *                         22 bits (absolute loss <= 2^-39 of that maximum).  inf / nan anywhere in a row or column
 *                         make the whole output row / column nan.  Runs for single (un-batched) products on 16-byte
 *                         aligned operands with K % 16 == 0; every other launch (batched attention products,
 *                         convolutions, epilogue column sums) takes the bf16 split: npm_last_math() tells which ran. */
enum { NPM_MATH_F32 = 0, NPM_MATH_BF16X3_FAST = 1, NPM_MATH_BF16X3 = 2, NPM_MATH_F16X2 = 3 };
/* The parity contract of a mode: what a layer computed under it may differ from the NumPy reference evaluated in fp64 on
 * the same inputs, on every configuration of BASELINE.json (C1 .. C5) at full width.  north_star's bound is "1e-4 rel";
 * fp32 contractions reorder sums, so it is read as
 *   REL     max |got - ref| / |ref|       over the elements with |ref| >= 0.1 max |ref|   (elementwise relative), and
 *   SCALED  max |got - ref| / max |ref|   over all elements                               (tensor-scaled).
 * tests/test_gpu_parity.py asserts both for every mode that has a line here (np_modeling_amd/_C.py reads the numbers from
 * this header), and bench.py reports a throughput line only for such a mode: a mode that fails its bound on any
 * configuration loses its line.  Measured worst cases (profiles/r06_parity_relative_error.log): f32 6.2e-5 / 6.7e-6,
 * bf16x3 1.9e-5 / 2.0e-6, f16x2 2.0e-5 / 2.1e-6 -- the weight gradient dwq of C4 each time.  NPM_MATH_BF16X3_FAST has no
 * line: its documented accumulation bias is outside the contract and bench.py does not report it. */
#define NPM_PARITY_REL_F32        1e-4
#define NPM_PARITY_SCALED_F32     1e-5
#define NPM_PARITY_REL_BF16X3     1e-4
#define NPM_PARITY_SCALED_BF16X3  1e-5
#define NPM_PARITY_REL_F16X2      1e-4
#define NPM_PARITY_SCALED_F16X2   1e-5
int npm_set_math(int mode);
int npm_get_math(void);                  /* the mode REQUESTED with npm_set_math */
/* The mode the most recent npm_sgemm / npm_conv2d_* / npm_mha_core_* launch actually RAN.  The split-bf16 modes exist
 * on the LDS-DMA pipelines with the 128 x 128 tile; a launch that cannot take them (operands not 16-byte aligned, K not a
 * multiple of 16, the epilogue column sums `colsum`, the 128 x 256 tile, the fused attention core) runs the exact-f32
 * MFMA and says so here -- results are then bit-equal to NPM_MATH_F32. */
int npm_last_math(void);
/* Diagnostics: when buf != NULL every block of the LDS-DMA GEMM writes 8 words (hardware id, XCC id, s_memtime at
 * start / first tile landed / loop end / after the epilogue stores) to buf[blockIdx*8 ..]; NULL switches it off. */
int npm_debug_gemm_trace(long long *buf);

/* ---- elementwise ---------------------------------------------------------- */
int npm_relu_fwd(const float *x, float *y, size_t n);                      /* activations.py:15 */
int npm_relu_bwd(const float *x_pre, const float *dy, float *dx, size_t n);/* activations.py:19 (x >= 0) */
int npm_add(const float *a, const float *b, float *out, size_t n);          /* transformer.py:39,53,78,90 */
int npm_add3(const float *a, const float *b, const float *c, float *out, size_t n); /* transformer.py:85 */
int npm_axpy(float *y, const float *x, float alpha, size_t n);              /* y += alpha*x; optimizer.py:32 */
int npm_scale(const float *x, float *y, float alpha, size_t n);
int npm_colsum(const float *x, float *out, int64_t rows, int64_t cols, int64_t ld); /* mlp.py:34 */
/* dx = (x_pre >= 0 ? dy : 0) on a contiguous [rows, cols] matrix and colsum[c] = sum_r dx[r, c] in the same
 * pass: the ReLU backward and the bias gradient of conv.py:54-55 / mlp.py:34,74 (one read of x_pre and dy). */
int npm_relu_bwd_colsum(const float *x_pre, const float *dy, float *dx, float *colsum, int64_t rows, int64_t cols);

/* ---- row kernels (one wavefront per row) ---------------------------------- */
/* out[(b*H + h)*S + s] = sum_d a[b,s,h,d] * b[b,s,h,d]: the row term of the fused softmax backward */
int npm_attn_rowdot(const float *a, const float *b, float *out, int64_t batch, int64_t seq, int64_t heads, int64_t dim);
/* y = softmax(scale * x) over the last axis                    (activations.py:26-29, attentions.py:104) */
int npm_softmax_fwd(const float *x, float *y, int64_t rows, int64_t n, float scale);
/* dx = scale * y * (dy - sum(dy*y)): closed form of the Jacobian einsum (activations.py:32-45, attentions.py:155) */
int npm_softmax_bwd(const float *y, const float *dy, float *dx, int64_t rows, int64_t n, float scale);
/* z = gamma*(x-mean)*rstd + beta; saves mean and rstd = 1/sqrt(var+eps), biased var (normalizations.py:45-48) */
int npm_layernorm_fwd(const float *x, const float *gamma, const float *beta, float eps,
                      int64_t rows, int64_t d, float *z, float *mean, float *rstd);
/* dx = rstd*(g - mean(g) - yhat*mean(g*yhat)) [+ residual], g = dz*gamma;
 * dgamma = sum dz*yhat, dbeta = sum dz (normalizations.py:50-75) */
int npm_layernorm_bwd(const float *dz, const float *x, const float *mean, const float *rstd,
                      const float *gamma, const float *residual, int64_t rows, int64_t d,
                      float *dx, float *dgamma, float *dbeta);
/* LayerNormalization of DropOut's output without ever storing it -- in the reference's encoder / decoder a DropOut always sits
 * directly in front of a LayerNormalization (transformer.py:35-36,40-41,49-50,55-56):
 *   forward   xd = mask ? x / keep_prob : 0 (normalizations.py:21-23) on the way in, then npm_layernorm_fwd's arithmetic on xd;
 *   backward  the same xd again from x and mask, npm_layernorm_bwd's arithmetic, then DropOut.backward on the way out
 *             (normalizations.py:27-30): dx = mask ? dxd / keep_prob : 0 [+ residual].
 * mask: one byte per element (0 = dropped), 4-byte aligned.  Equal to npm_mask_scale + npm_layernorm_fwd / npm_layernorm_bwd +
 * npm_mask_scale (+ npm_add) to rounding (the same operations in the same order; the compiler contracts the row sums of products
 * into fused multiply-adds per kernel instance: single ulps in a few per cent of dx).  Rows with d % 4 != 0 or d > 4096 are NPM_E_UNSUPPORTED: compose the calls above. */
int npm_layernorm_dropout_fwd(const float *x, const unsigned char *mask, float keep_prob, const float *gamma, const float *beta,
                              float eps, int64_t rows, int64_t d, float *z, float *mean, float *rstd);
int npm_layernorm_dropout_bwd(const float *dz, const float *x, const unsigned char *mask, float keep_prob, const float *mean,
                              const float *rstd, const float *gamma, const float *residual, int64_t rows, int64_t d,
                              float *dx, float *dgamma, float *dbeta);

/* ---- Conv2D: NHWC x HWIO, SAME, stride 1, odd k (layers/conv.py:74-194) ----
 * Implicit-im2col GEMM on the fp32 MFMA; the im2col matrix is never materialised. */
typedef struct npm_conv2d {
    int32_t n, h, w, c_in, c_out, ksize;
    const float *x;        /* [n,h,w,c_in] */
    const float *filt;     /* [k,k,c_in,c_out] */
    const float *bias;     /* [c_out] or NULL */
    float *y;              /* [n,h,w,c_out] */
    float *pre;            /* optional pre-activation output when relu != 0 */
    int32_t relu;
} npm_conv2d;
int npm_conv2d_fwd(const npm_conv2d *c);                                           /* conv.py:44-48,97-105 */
/* dx = conv(dy, flip+transpose(filt))  (conv.py:130,153) */
int npm_conv2d_bwd_x(const float *dy, const float *filt, float *dx,
                     int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize);
/* dw[i,j] = shifted(x)^T dy  (conv.py:185-194) */
int npm_conv2d_bwd_w(const float *dy, const float *x, float *dw,
                     int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize);
/* Conv2D.backward's first three lines in one call (conv.py:54-56 with the layer's ReLU, activations.py:19):
 *   g = where(pre >= 0, dy, 0)  [n,h,w,c_out], written out (the grad_x convolution reads it: npm_conv2d_bwd_x(g, ...))
 *   db = sum over n, h, w of g  [c_out]
 *   dw[i,j] = shifted(x)^T g    [k,k,c_in,c_out]
 * The mask is applied where the grad_w kernel stages its dy tiles, so no separate ReLU-backward pass over the
 * activation-sized tensors runs; results are bitwise reproducible (fixed-order slab and column reductions). */
int npm_conv2d_bwd_w_relu(const float *dy, const float *pre, const float *x, float *g, float *dw, float *db,
                          int32_t n, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t ksize);

/* ---- fused attention core (layers/attentions.py:103-112 forward, :146-162 backward) ----
 * ctx[b, i, h, :] = sum_j softmax_j(scale * q[b, i, h, :] . k[b, j, h, :]) v[b, j, h, :] in ONE kernel: the
 * [B, H, Sq, Skv] probabilities never go to memory (online softmax, as derived in the reference's
 * layers/attentions_test.py:158-265); the forward saves lse[b, h, i] = log sum_j exp(scale * q.k) and the backward
 * recomputes the probabilities from it.  q/k/v/ctx and the gradients are [B, S, H, D] with a row pitch (H * D, or
 * 3 * H * D inside a packed qkv buffer); head_dim in {16, 32, 64, 128}, else NPM_E_UNSUPPORTED (callers then
 * compose the same math from npm_sgemm + npm_softmax_*).  mask (optional): bytes, element (b, h, i, j) at
 * mask[b * stride_b + h * stride_h + i * stride_q + j], 0 = excluded (np.where(mask, scaled, -inf),
 * attentions.py:105-107); the backward treats excluded positions as P = 0 (the reference raises NotImplementedError
 * there, attentions.py:152-153).  scores (optional, [B, H, Sq, Skv]): when given, the forward also stores the raw
 * masked scores q.k and the backward reads them instead of recomputing q.k (trades 4 B/element of traffic each way
 * for one of the five matrix products).  fp32 MFMA only (npm_set_math does not apply to this kernel). */
typedef struct npm_mha_core {
    int32_t batch, heads, seq_q, seq_kv, head_dim;
    float scale;                                   /* 1 / sqrt(Dk), > 0 */
    const float *q; int64_t q_pitch;
    const float *k; int64_t k_pitch;
    const float *v; int64_t v_pitch;
    const uint8_t *mask; int64_t mask_stride_b, mask_stride_h, mask_stride_q;
    float *ctx; int64_t ctx_pitch;                 /* forward: out; backward: in (the forward's result) */
    float *lse;                                    /* [B, H, Sq]; forward: out; backward: in */
    float *scores;                                 /* optional [B, H, Sq, Skv]; forward: out; backward: in */
    const float *dctx; int64_t dctx_pitch;         /* backward only from here */
    float *dq; int64_t dq_pitch;
    float *dk; int64_t dk_pitch;
    float *dv; int64_t dv_pitch;
    /* Optional, with `mask`: the mask's tile summary from npm_mha_mask_summary -- one byte per (query tile of 32, key block
     * of 128), bit w set when some position of the 32 x 16 sub-tile (keys 16 w .. 16 w + 15 of the block) is allowed; byte
     * (qt, kb) of plane (b, h) at tile_summary[b * summary_stride_b + h * summary_stride_h + qt * ceil(seq_kv / 128) + kb]
     * (a stride of 0 broadcasts, like the mask's).  Forward and backward then skip tiles without an allowed position: the
     * results are the same as without it -- including the NaNs of a query row with NO allowed key (np.where(mask, s, -inf) then
     * softmax: that row of ctx and dq, and through P = NaN every dk / dv row of its (batch, head)): npm_mha_mask_summary marks
     * every tile of a mask plane that has such a row as "visit", so such planes are simply not skipped.  Positions of `scores`
     * inside skipped tiles are left unwritten (the backward never reads them).  Used for seq_q, seq_kv <= 2048; longer
     * sequences, and calls made while npm_debug_attn_trace is active, run unskipped. */
    const uint8_t *tile_summary; int64_t summary_stride_b, summary_stride_h;
    int64_t summary_all_offset;   /* bytes from a tile's "some position allowed" byte to its "every position allowed" byte (the second
                                     half of what npm_mha_mask_summary writes: planes_b * planes_h * tiles bytes later); 0 = not given.
                                     Tiles whose every position is allowed run without reading the mask. */
    /* Optional, backward: the row terms MINUS scale * (dctx_i . ctx_i) already computed by the caller (the NPM_EPI_ROWDOT
     * epilogue of the GEMM that produced dctx -- which exists at head_dim 128 only, but any head size is taken here), element
     * (b, h, i) at neg_delta[b * stride_b + h * stride_h + i], 16-byte aligned, strides multiples of 4; npm_mha_core_bwd then
     * does not read dctx and ctx for them.  Honoured by the eight-wave kernels (mha_bwd16_kernel, mha_bwd8_kernel) when
     * seq_q % 4 == 0 (they fetch four row terms per load); otherwise -- and on the four-wave kernels (NPM_TUNE_ATTN_BWD16 = 0
     * or an active npm_debug_attn_trace) -- it is ignored and the terms are recomputed from dctx and ctx: same results. */
    const float *neg_delta; int64_t neg_delta_stride_b, neg_delta_stride_h;
} npm_mha_core;
int npm_mha_core_supported(int head_dim);          /* 1 when npm_mha_core_fwd/bwd take this head dimension */
int npm_mha_core_fwd(const npm_mha_core *c);
int npm_mha_core_bwd(const npm_mha_core *c);
/* A plane (b, h) with a query row that has no allowed key at all gets 0xFF in every "some position allowed" byte (see above).
 * summary[2][plane_b][plane_h][ceil(seq_q / 32)][ceil(seq_kv / 128)] (first the "some position allowed" bytes, then, in the
 * same order, the "every position inside the tensors allowed" bytes) of a byte mask laid out like npm_mha_core's (element
 * (b, h, i, j) at mask[b * stride_b + h * stride_h + i * stride_q + j]); planes_b / planes_h = how many distinct planes the
 * mask has along batch and head (1 where it broadcasts).  np.where(mask, scaled, -inf) of attentions.py:105-107 skips
 * nothing; this is what lets the fused kernels skip the tiles such a mask empties (half of a causal mask's). */
int npm_mha_mask_summary(const uint8_t *mask, int64_t stride_b, int64_t stride_h, int64_t stride_q, int32_t planes_b,
                         int32_t planes_h, int32_t seq_q, int32_t seq_kv, uint8_t *summary);
/* Which kernel the most recent npm_mha_core_fwd / npm_mha_core_bwd call launched, as "<kernel> D=<head_dim> mask=<0|1>
 * scores=<0|1>" (e.g. "mha_bwd16_kernel D=128 mask=0 scores=1"); "" before the first call.  Tests use it to assert that
 * a comparison exercised the kernel it names. */
const char *npm_last_attn_kernel(void);
/* Diagnostics: when buf != NULL every block of the backward kernel writes 16 words of s_memtime stamps of ONE of its
 * tiles (phase boundaries: tile start, after S, dP, dV, dK, the dS barrier, dQ, the dQ stores; word 8: next tile's start)
 * to buf[blockIdx * 16 ..]; NULL switches it off. */
int npm_debug_attn_trace(long long *buf);

/* ---- around the path ("next" rows of SURVEY.md section 8f): keeps a Trainer step on the device ---- */
/* Adam with the reference's numerics (optimizer.py:53-67): fp64 moments m, v (device buffers of n doubles,
 * zero-initialised with npm_fill_f64), bias correction with step >= 1, epsilon inside the sqrt */
int npm_adam_step(float *var, const float *grad, double *m, double *v, size_t n, double lr, double beta1,
                  double beta2, double eps, int step);
int npm_fill_f64(double *dst, double value, size_t n);
/* MSELoss (loss.py:21-29): loss = sum((y-t)^2)/n (fp64 accumulation, returned to the host); dy = 2 (y-t) / n */
int npm_mse_fwd(const float *y, const float *targets, size_t n, double *loss);
int npm_mse_bwd(const float *y, const float *targets, float *dy, size_t n);
/* CrossEntropyLoss (loss.py:33-39): loss = -sum(t log y); dy = -t / y */
int npm_xent_fwd(const float *y, const float *targets, size_t n, double *loss);
int npm_xent_bwd(const float *y, const float *targets, float *dy, size_t n);
/* DropOut (normalizations.py:14-30): y = mask ? x / keep_prob : 0 with a host-drawn byte mask */
int npm_mask_scale(const float *x, const unsigned char *mask, float *y, size_t n, float keep_prob);
/* The same with the mask drawn ON THE DEVICE: element i keeps its value when word (i & 3) of
 * Philox4x32-10(counter = (i / 4, offset), key = seed) is below keep_prob * 2^32; the byte mask is written too (the layer's
 * `_mask` stays readable).  Deterministic in (seed, offset); not the host generator's stream -- seeded parity with the
 * reference needs the host-drawn path above.  x == y == NULL: only the mask is drawn (its consumer applies it:
 * npm_layernorm_dropout_fwd / _bwd). */
int npm_dropout_philox(const float *x, float *y, unsigned char *mask, size_t n, float keep_prob, uint64_t seed, uint64_t offset);

#ifdef __cplusplus
}
#endif
#endif /* NPM_HIP_H */
