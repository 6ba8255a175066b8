/* vitpe.h -- C ABI of libvitpe.so: the MI355X (gfx950) kernels behind the ViT
 * attention-with-positional-encoding hot path of zhengyk19/vit-rpe-rope.
 *
 * The reference is pure Python on ATen ops and has no FFI of its own; the boundary it
 * exposes is the Python API models.vit.VisionTransformer(...) + --pos_encoding (reference
 * models/vit.py:148-151, train.py:33-34).  Each entry point below replaces the ATen op
 * sequence of the cited reference lines; the Python host (vit-rpe-rope_amd/vitpe) binds
 * them with ctypes and registers them as torch.library custom ops.
 *
 * Conventions
 *  - every function returns a hipError_t as int (0 = success); nothing throws, nothing
 *    synchronises, nothing allocates: the caller owns every buffer (device pointers);
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); re-entrant, no global
 *    state, capturable into a hipGraph;
 *  - `dtype` selects the arithmetic/storage type T of activations and GEMM weights:
 *    VITPE_F32 (exact fp32 MFMA, the 1e-4 parity mode) or VITPE_BF16 (bf16 MFMA, fp32
 *    accumulate).  LayerNorm/bias/positional parameters, statistics, logits, gradients of
 *    parameters and optimizer state are always fp32;
 *  - matrices are dense row-major.
 */
#ifndef VITPE_H
#define VITPE_H

#ifdef __cplusplus
extern "C" {
#endif

#define VITPE_F32 0
#define VITPE_BF16 1

/* --pos_encoding switch, reference train.py:33-34 / models/vit.py:170-196 */
#define VITPE_PE_NONE 0
#define VITPE_PE_ABSOLUTE 1
#define VITPE_PE_RELATIVE 2
#define VITPE_PE_POLYNOMIAL 3
#define VITPE_PE_ROPE_AXIAL 4
#define VITPE_PE_ROPE_MIXED 5

/* gemm_nt epilogues */
#define VITPE_EPI_BIAS 0      /* C = A W^T + bias                       (attn.qkv / any Linear)        */
#define VITPE_EPI_BIAS_GELU 1 /* U = A W^T + bias ; C = gelu_erf(U)     (timm Mlp fc1+act, vit.py:118) */
#define VITPE_EPI_BIAS_RESID 2/* C = A W^T + bias + R                   (attn.proj / fc2 + residual, vit.py:91,122,124) */
#define VITPE_EPI_PATCH 3     /* patch-embed: row remap + cls row + APE (vit.py:248-258)               */
#define VITPE_EPI_GELU_BWD 4  /* C = (A W^T) * gelu'(U)                 (autograd of fc1's GELU)       */

typedef void* vitpe_stream_t; /* hipStream_t */

int vitpe_abi_version(void);

/* ---- fused attention (the north-star op) --------------------------------------------------
 * Replaces reference models/vit.py:47-88 (qkv Linear, head split, RoPE rotate-half on patch
 * tokens via rope_utils.py:18-37, QK^T*hd^-0.5, + RelativePositionalEncoding /
 * PolynomialRPE bias (positional_encoding.py:82-95,127-171), softmax, @V, head merge).
 *   xn    [B,N,D] T   layer-normed tokens (N = patches+1, class token first)
 *   wqkv  [3D,D]  T   attn.qkv.weight (no bias: Block passes qkv_bias=False, vit.py:110,200),
 *                     PACKED fragment-major by vitpe_pack_qkv_weights (same element count)
 *   out   [B,N,D] T   merged heads, input of attn.proj
 *   cos/sin: rope-axial [P,HD/2], rope-mixed [H,P,HD/2] contiguous fp32 (else NULL)
 *   table : relative [H,2N-1] fp32 ; coeff: polynomial [deg+1] or [H,deg+1] fp32 (else NULL)
 *   grid  : patches per side (sqrt(P)); degree <= 7.
 * Supported shapes: vitpe_fused_attention_supported() (HD=32, 65<=N<=80, D in {96,192});
 * anything else returns hipErrorNotSupported.                                              */
int vitpe_fused_attention_supported(int dtype, int N, int D, int HD);
/* fp32 master [3D,D] -> T, re-ordered so that each MFMA weight fragment of a wave is one contiguous
 * 1 KB read: block (head, {q,k,v}, 16-row tile, 32-deep K chunk) x 64 lanes x 8 elements          */
int vitpe_pack_qkv_weights(int dtype, const float* wqkv, void* packed, int D, int HD, vitpe_stream_t stream);
int vitpe_fused_attention_fwd(int dtype, const void* xn, const void* wqkv, void* out, int B, int N,
                              int D, int HD, int mode, const float* cos, const float* sin,
                              const float* table, const float* coeff, int grid, int degree,
                              int coeff_per_head, vitpe_stream_t stream);
/* Same with the preceding LayerNorm fused into the token staging: x = RAW tokens, mean/rstd their
 * row statistics, xn_out (nullable) receives LayerNorm(x) (needed by the backward pass).          */
int vitpe_fused_attention_fwd_ln(int dtype, const void* x, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, void* xn_out, const void* wqkv,
                                 void* out, int B, int N, int D, int HD, int mode, const float* cos,
                                 const float* sin, const float* table, const float* coeff, int grid,
                                 int degree, int coeff_per_head, vitpe_stream_t stream);
/* ---- the "wide" forward: 32x32x16 matrix-core tiles for the benchmark geometry (bf16, N = 65, D = 192, HD = 32) ------
 * Same operation as vitpe_fused_attention_fwd(_ln) (reference models/vit.py:47-88, LayerNorm vit.py:113,122 optional),
 * on weights packed by vitpe_pack_qkv_weights_wide: vitpe_qkv_wide_pack_elems(D) = 3 D D elements of T, the
 * 32x32x16 operand fragments (block (head, {q,k,v}, 16-deep k step) x 64 lanes x 8 = 1 KB, what one LDS-DMA
 * instruction moves); the q rows are pre-multiplied by HD^-0.5 * log2(e).
 * gamma == NULL: x is already layer-normed (beta / mean / rstd / xn_out ignored).  Other shapes: hipErrorNotSupported. */
int vitpe_fused_attention_wide_supported(int dtype, int N, int D, int HD);
int vitpe_qkv_wide_pack_elems(int D);
int vitpe_pack_qkv_weights_wide(int dtype, const float* wqkv, void* packed, int D, int HD, vitpe_stream_t stream);
int vitpe_fused_attention_fwd_wide(int dtype, const void* x, const float* gamma, const float* beta,
                                   const float* mean, const float* rstd, void* xn_out, const void* wqkv_wide,
                                   void* out, int B, int N, int D, int HD, int mode, const float* cos,
                                   const float* sin, const float* table, const float* coeff, int grid, int degree,
                                   int coeff_per_head, vitpe_stream_t stream);
/* Backward of the above (the reference relies on autograd).  dqkv [B,N,3D] T is the gradient of
 * the qkv Linear's output (columns [q|k|v] x heads, like the forward's qkv buffer); the caller
 * finishes with dxn = dqkv Wqkv (vitpe_gemm_nt on the transposed shadow) and dWqkv = dqkv^T xn
 * (vitpe_gemm_tn).  dtable / dcoeff / dfreqs are ACCUMULATED into (fp32 atomics).           */
int vitpe_fused_attention_bwd(int dtype, const void* xn, const void* wqkv, const void* dout,
                              void* dqkv, int B, int N, int D, int HD, int mode, const float* cos,
                              const float* sin, const float* table, const float* coeff, int grid,
                              int degree, int coeff_per_head, float* dtable, float* dcoeff,
                              float* dfreqs, vitpe_stream_t stream);
/* The same with the LayerNorm recomputed while staging: x = RAW tokens, mean / rstd their row statistics (as
 * vitpe_fused_attention_fwd_ln) -- the forward then need not store LayerNorm(x) at all.                       */
int vitpe_fused_attention_bwd_ln(int dtype, const void* x, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, const void* wqkv, const void* dout,
                                 void* dqkv, int B, int N, int D, int HD, int mode, const float* cos,
                                 const float* sin, const float* table, const float* coeff, int grid,
                                 int degree, int coeff_per_head, float* dtable, float* dcoeff,
                                 float* dfreqs, vitpe_stream_t stream);

/* ---- attention core on a qkv buffer (geometries the fused kernels do not cover) -----------
 * Replaces models/vit.py:49-92 between `qkv = self.qkv(x)` and `self.proj`: head split, RoPE on
 * q,k (rope_utils.py:85-101; class token not rotated), QK^T*hd^-0.5 + relative / polynomial bias
 * (positional_encoding.py:82-95,127-171), softmax, @V, head merge.  One workgroup per (image, head).
 *   qkv  [B,N,3*H*HD] T  output of the qkv Linear (vitpe_linear), columns [q|k|v] x heads x HD
 *   out  [B,N,H*HD]   T  merged heads ; dqkv same layout as qkv
 * PE operands and gradient accumulation as for vitpe_fused_attention_*.  Supported:
 * vitpe_attention_core_supported() -- (HD=64, 193<=N<=208: 224x224 / patch 16) and (HD=32, 65<=N<=80);
 * rope-mixed needs H <= 16.  Anything else returns hipErrorNotSupported.                         */
int vitpe_attention_core_supported(int dtype, int N, int HD);
int vitpe_attention_core_fwd(int dtype, const void* qkv, void* out, int B, int N, int H, int HD, int mode,
                             const float* cos, const float* sin, const float* table, const float* coeff,
                             int grid, int degree, int coeff_per_head, vitpe_stream_t stream);
int vitpe_attention_core_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, int B, int N, int H,
                             int HD, int mode, const float* cos, const float* sin, const float* table,
                             const float* coeff, int grid, int degree, int coeff_per_head, float* dtable,
                             float* dcoeff, float* dfreqs, vitpe_stream_t stream);
/* vitpe_attention_fused64_fwd: the reference's Attention.forward before self.proj (models/vit.py:47-88) at the ViT-B/16
 * geometry (BASELINE config 5: hd = 64, N = 197) as ONE kernel -- the head's slice of the qkv projection, the rotation /
 * bias, QK^T, softmax and .V per (image, head); q and k never leave the chip.  xn [B,N,D] T = LayerNorm1's output
 * (D = 64 H); wqkv_packed = vitpe_pack_weight_frags(attn.qkv.weight [3D,D], kchunk 64, phi 0); qkv_out (nullable)
 * [B,N,3D] T receives the raw projection, which is what vitpe_attention_core_bwd reads in training; out [B,N,D] T merged
 * heads.  bf16, hd = 64, 193 <= N <= 208, H <= 16 (vitpe_attention_fused64_supported); else hipErrorNotSupported: run
 * vitpe_linear + vitpe_attention_core_fwd.  PE arguments as vitpe_attention_core_fwd.                                   */
int vitpe_attention_fused64_supported(int dtype, int N, int H, int HD);
int vitpe_attention_fused64_fwd(int dtype, const void* xn, const void* wqkv_packed, void* qkv_out, void* out, int B,
                                int N, int H, int HD, int mode, const float* cos, const float* sin,
                                const float* table, const float* coeff, int grid, int degree, int coeff_per_head,
                                vitpe_stream_t stream);

/* ---- GEMMs ------------------------------------------------------------------------------
 * vitpe_gemm_nt: C[M,N] = epi(A[M,K] W[N,K]^T).  nn.Linear forward (vit.py:35,37; timm Mlp
 * fc1/fc2), nn.Conv2d-as-GEMM patch embed (vit.py:164,248), and -- on a transposed weight
 * shadow -- the data gradients.  N % 8 == 0; K % 8 == 0 (bf16) / K % 4 == 0 (fp32).
 *   bias [N] fp32 or NULL; R residual [M,N] T (EPI_BIAS_RESID); U [M,N] T pre-activation
 *   (written by EPI_BIAS_GELU, read by EPI_GELU_BWD); EPI_PATCH: A = unfolded patches
 *   [B*P,K], C = tokens [B*(P+1),N], ape [P,N] fp32 or NULL, cls [N] fp32.                 */
int vitpe_gemm_nt(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                  const void* R, void* U, const float* ape, const float* cls, int M, int N, int K,
                  int P, int Ntok, vitpe_stream_t stream);
/* vitpe_linear: same contract as vitpe_gemm_nt (minus EPI_PATCH) on the second-generation panel
 * kernel (144x192 tiles = full output rows for N = 192; used whenever N % 192 == 0, otherwise it
 * forwards to vitpe_gemm_nt).  mean_out/rstd_out (both or neither; N == 192 only) receive the
 * LayerNorm statistics of the OUTPUT rows (eps as in nn.LayerNorm), so the following LayerNorm
 * (vit.py:113,116) needs no pass of its own.                                                 */
int vitpe_linear(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                 const void* R, void* U, float* mean_out, float* rstd_out, float eps, int M, int N,
                 int K, vitpe_stream_t stream);
/* LayerNorm fused into its neighbours (removes the stand-alone LayerNorm passes of vit.py:113,116):
 *  vitpe_linear_ln    : C = epi(LN(X) W^T), X raw [M,K] with row statistics mean/rstd (e.g. the stats
 *                       output of vitpe_linear); xn_out (nullable) receives LN(X) for the backward pass;
 *                       epi in {BIAS, BIAS_GELU}, N % 192 == 0.
 *  vitpe_linear_lnbwd : dx = dres + LN'(dY Wt^T) for the LayerNorm with input rows x [M,192]; dgamma/dbeta
 *                       accumulated.  (data gradient of qkv / fc1 + LayerNorm backward + residual add)      */
int vitpe_linear_ln(int dtype, int epi, const void* X, const float* gamma, const float* beta,
                    const float* mean, const float* rstd, void* xn_out, const void* W, void* C,
                    const float* bias, void* U, int M, int N, int K, vitpe_stream_t stream);
int vitpe_linear_lnbwd(int dtype, const void* dY, const void* Wt, void* dx, const void* x, const float* mean,
                       const float* rstd, const float* gamma, const void* dres, float* dgamma, float* dbeta,
                       int M, int K, vitpe_stream_t stream);
/* vitpe_linear_lnbwd2: the same function on the wave-per-tile mapping of vitpe_block_tail2_*, with the weight given as
 * Wt_packed = vitpe_pack_weight_frags(W^T [192,K], kchunk 64, phi 0), W = the Linear's weight [K,192] (attn.qkv.weight:
 * K = 576).  bf16, K % 192 == 0; otherwise hipErrorNotSupported.                                                        */
int vitpe_linear_lnbwd2(int dtype, const void* dY, const void* Wt_packed, void* dx, const void* x, const float* mean,
                        const float* rstd, const float* gamma, const void* dres, float* dgamma, float* dbeta, int M,
                        int K, vitpe_stream_t stream);
/* vitpe_block_tail2_fwd: the attention branch's tail and the MLP branch of a block in one kernel (vit.py:91,116-118,
 * 122-124 with timm Mlp):
 *   x_mid = x_in + attn_out Wp^T + bp ;  xn = LayerNorm2(x_mid) ;  u = xn W1^T + b1 ;  h = gelu(u) ;  out = x_mid + h W2^T + b2
 * x_mid [M,192] and its LayerNorm statistics mean2 / rstd2 [M] are outputs (backward needs them); xn_out (nullable) is kept
 * for fc1's weight gradient; mean_out / rstd_out (both or neither) receive the statistics of the OUTPUT rows (the next
 * block's LayerNorm1; eps2: norm2, eps_next: those).  Each wave carries one 16-token tile through the whole chain (hidden
 * activation in registers between fc1 and fc2) and reads the weights as MFMA fragments from fragment-major packed copies
 * made by vitpe_pack_weight_frags:
 *   Wp_packed = pack(attn.proj.weight [192,192], kchunk 192, phi 0)
 *   W1_packed = pack(mlp.fc1.weight  [HID,192], kchunk 192, phi 1)
 *   W2_packed = pack(mlp.fc2.weight  [192,HID], kchunk  32, phi 1)
 * What it keeps of the hidden layer for backward is h_out = gelu(u) (bf16) and gp_out = gelu'(u) [M,HID] as IEEE HALF
 * (2 bytes like bf16, 11 significant bits: vitpe_block_tail2_bwd multiplies every du element by it) -- NOT u: the
 * backward multiplies by the stored derivative; pass both or, for inference, neither.  bf16, D = 192, HID % 64 == 0, 128 <= HID <= 1536 (vitpe_block_tail2_supported); otherwise
 * hipErrorNotSupported -- run vitpe_linear / vitpe_linear_ln instead.
 * vitpe_pack_weight_frags: W fp32 [R,C] (R % 16 == 0, kchunk % 32 == 0, C % kchunk == 0) -> 1-KB fragments
 * (64 lanes x 8 elements) at fragment index ((kc * R/16 + nt) * kchunk/32 + ks); lane 16g + cc, element e holds
 * W[16nt + cc][kchunk*kc + 32ks + k], k = 8g + e (phi 0) or (e < 4 ? 4g + e : 16 + 4g + e - 4) (phi 1).          */
int vitpe_pack_weight_frags(int dtype, const float* W, void* packed, int R, int C, int kchunk, int phi,
                            vitpe_stream_t stream);
int vitpe_block_tail2_supported(int dtype, int D, int HID);
int vitpe_block_tail2_fwd(int dtype, const void* attn_out, const void* x_in, const void* Wp_packed, const float* bp,
                          const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                          void* xn_out, const void* W1_packed, const float* b1, const void* W2_packed,
                          const float* b2, void* gp_out, void* h_out, void* out, float* mean_out, float* rstd_out,
                          float eps2, float eps_next, int M, int D, int HID, vitpe_stream_t stream);
/* vitpe_block_tail2_bwd: backward of vitpe_block_tail2_fwd w.r.t. its inputs, the same pipeline on the transposes:
 *   du = (dy fc2.weight) * gp  [M,HID] (stored: fc1's weight gradient reads it),  dx_mid = dy + LayerNorm2'(du fc1.weight)
 *   da = dx_mid attn.proj.weight  [M,192] (input of the attention backward);  dgamma / dbeta of norm2 accumulated (fp32
 *   atomics, one per column and workgroup).  gp = gelu'(u) as saved by vitpe_block_tail2_fwd (IEEE half); x_mid / mean2 / rstd2 its
 *   LayerNorm2 input rows and statistics.  Weights as vitpe_pack_weight_frags copies of the TRANSPOSES:
 *   W2t_packed = pack(fc2.weight^T [HID,192], kchunk 192, phi 1), W1t_packed = pack(fc1.weight^T [192,HID], kchunk 32, phi 1),
 *   WpT_packed = pack(attn.proj.weight^T [192,192], kchunk 192, phi 1).  Support as vitpe_block_tail2_fwd.                 */
int vitpe_block_tail2_bwd(int dtype, const void* dy, const void* gp, const void* W2t_packed, const void* W1t_packed,
                          const void* x_mid, const float* mean2, const float* rstd2, const float* gamma, void* du,
                          void* dx_mid, float* dgamma, float* dbeta, const void* WpT_packed, void* da, int M, int D,
                          int HID, vitpe_stream_t stream);
/* vitpe_block_tail2_bwd_pre: the same preceded, in the same kernel, by the qkv data gradient + LayerNorm1 backward + residual
 * of the block ABOVE (vitpe_linear_lnbwd2's function):  dy_out = dres1 + LayerNorm1'(d_qkv Wqkv)  is written (the weight
 * gradients read it) and feeds the MLP backward from registers -- one launch, one HBM round trip of dy and one kernel
 * prologue less per layer.  WqT_packed = pack(attn.qkv.weight^T [192,K1], kchunk 64, phi 0) of the upper block; x1 / mean1 /
 * rstd1 / gamma1: its LayerNorm1 input rows, statistics and weight; dgamma1 / dbeta1 accumulated.  K1 % 64 == 0, K1 <= 640. */
int vitpe_block_tail2_bwd_pre(int dtype, const void* d_qkv, const void* WqT_packed, const void* x1, const float* mean1,
                              const float* rstd1, const float* gamma1, const void* dres1, float* dgamma1, float* dbeta1,
                              int K1, void* dy_out, const void* gp, const void* W2t_packed, const void* W1t_packed,
                              const void* x_mid, const float* mean2, const float* rstd2, const float* gamma, void* du,
                              void* dx_mid, float* dgamma, float* dbeta, const void* WpT_packed, void* da, int M, int D,
                              int HID, vitpe_stream_t stream);
/* vitpe_gemm_tn: dW[N,K] += dY[M,N]^T X[M,K] ; dbias[N] += colsum(dY) (NULL to skip).  fp32
 * outputs, accumulated with atomics over `splits` token slices.                             */
int vitpe_gemm_tn(int dtype, const void* dY, const void* X, float* dW, float* dbias, int M, int N,
                  int K, int splits, vitpe_stream_t stream);

/* vitpe_wgrad_group: the same product for a LIST of independent problems (<= 28) in one launch --
 * e.g. every nn.Linear weight/bias gradient of the model (autograd's addmm backward in the
 * reference: vit.py:35,37 and timm Mlp fc1/fc2, once per block).  `problems` is a HOST array; the
 * descriptors travel as kernel arguments (no device table, graph-capture safe).  Work is cut into
 * (192x192 output block, 64-token stage) units spread evenly over the CUs, so each output block is
 * accumulated (fp32 atomics) by only a handful of workgroups.  N, K multiples of 8 (bf16) / 4 (fp32).  Lists whose every
 * problem has N % 192 == 0 and K % 384 == 0 (bf16, x_op 0: the ViT-B/16 shapes) run on 192 x 384 blocks instead.  */
typedef struct {
  const void* dY; /* [M,N] T */
  const void* X;  /* [M,K] T */
  float* dW;      /* [N,K] fp32, accumulated into */
  float* dbias;   /* [N] fp32 or NULL, accumulated into */
  int M, N, K;
  int x_op;       /* VITPE_XOP_NONE, or VITPE_XOP_LAYERNORM: the operand is LayerNorm(X) with the row statistics and
                   * affine parameters below, recomputed while staging (the normalised tensor is never stored: it is
                   * a pure function of X, which the forward keeps anyway, vit.py:122,124) */
  const float* x_mean;  /* [M] */
  const float* x_rstd;  /* [M] */
  const float* x_gamma; /* [K] */
  const float* x_beta;  /* [K] */
} vitpe_wgrad_problem;
#define VITPE_XOP_NONE 0
#define VITPE_XOP_LAYERNORM 1
int vitpe_wgrad_group(int dtype, const vitpe_wgrad_problem* problems, int nprob, vitpe_stream_t stream);

/* ---- LayerNorm (nn.LayerNorm(d), eps 1e-5: vit.py:113,116,210) ------------------------------ */
/* y == NULL: row statistics only (mean and rstd required) */
int vitpe_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, int M, int D, float eps, vitpe_stream_t stream);
int vitpe_layernorm_bwd_blocks(int M); /* workspace = blocks * 2 * D floats */
/* dx = (dres ? dres : 0) + LN'(dy): also folds the residual-branch gradient of vit.py:122,124 */
int vitpe_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean,
                        const float* rstd, const float* gamma, const void* dres, void* dx,
                        float* dgamma, float* dbeta, float* workspace, int M, int D,
                        vitpe_stream_t stream);
int vitpe_reduce_partials(const float* partial, int nparts, int len0, int len1, float* dst0,
                          float* dst1, vitpe_stream_t stream);

/* ---- patch embed (vit.py:164,245-258) ------------------------------------------------------- */
/* img [B,C,S,S] fp32 -> patches [B*P, C*p*p] T, column = c*p*p + ky*p + kx                    */
int vitpe_unfold(int dtype, const float* img, void* patches, int B, int C, int S, int p,
                 vitpe_stream_t stream);
/* The same from a uint8 dataset resident in HBM -- replaces the DataLoader gather + ToTensor + Normalize
 * of train.py:69-92 in front of the patch embed: record index[b] (NULL: record b) of data [Ndata,C,S,S]
 * uint8 is turned into ((x/255) - mean[c]) / std[c] (fp32, the reference's operation order) and written as
 * the patch matrix; img_out (nullable) receives the normalised fp32 image [B,C,S,S].                   */
int vitpe_unfold_u8(int dtype, const unsigned char* data, const long long* index, const float* mean,
                    const float* stdv, void* patches, float* img_out, int B, int C, int S, int p,
                    vitpe_stream_t stream);
/* Fused patch embedding for the small-K geometries (K = C p^2 <= 64, (S/p)^2 <= 64 patches, D <= 256, D % 16 == 0:
 * vitpe_patch_embed_supported): unfold (from fp32 images `img`, XOR from the resident uint8 dataset `data` + `index` with
 * ToTensor + Normalize as vitpe_unfold_u8) + Conv2d-as-GEMM + bias + absolute PE rows `ape` [P,D] (nullable) + class
 * token in row 0 -> tokens [B,P+1,D]; `patches` [B*P,K] (nullable) receives the patch matrix for the weight gradient;
 * mean / rstd (both or neither) the LayerNorm statistics of every token row (vit.py:113).  One launch instead of
 * vitpe_unfold + vitpe_gemm_nt(EPI_PATCH) + vitpe_layernorm_fwd.                                                   */
int vitpe_patch_embed_supported(int dtype, int C, int S, int p, int D);
int vitpe_patch_embed(int dtype, const float* img, const unsigned char* data, const long long* index,
                      const float* nmean, const float* nstd, const void* W, const float* bias, const float* cls,
                      const float* ape, void* tokens, void* patches, float* mean, float* rstd, int B, int C, int S,
                      int p, int D, float eps, vitpe_stream_t stream);
/* dcls[d] += sum_b dtok[b,0,d]; dape[p,d] += sum_b dtok[b,1+p,d] (NULL to skip);
 * dpatch [B*P,D] T = patch rows of dtok (input of the patch-embed weight gradient)            */
int vitpe_embed_bwd(int dtype, const void* dtok, float* dcls, float* dape, void* dpatch, int B,
                    int Ntok, int D, vitpe_stream_t stream);

/* ---- positional-encoding tables (models/positional_encoding.py) --------------------------- */
int vitpe_relative_position_index(long long* out, int L, vitpe_stream_t stream);   /* :67-75, int64 [L,L], bit-exact */
int vitpe_l1_distance_matrix(long long* out, int G, vitpe_stream_t stream);        /* :136-142, int64 [G*G,G*G], bit-exact */
int vitpe_rope_axial_tables(const float* inv_freq, float* cosv, float* sinv, int G, int half,
                            vitpe_stream_t stream);                                /* :228-245 */
int vitpe_rope_mixed_tables(const float* freqs, float* cosv, float* sinv, int H, int G, int half,
                            vitpe_stream_t stream);                                /* :325-351 incl. the view-scramble */
int vitpe_relative_bias(const float* table, float* out, int H, int L, vitpe_stream_t stream); /* :82-95 */
int vitpe_polynomial_bias(const float* coeff, float* out, int H, int G, int degree, int per_head,
                          vitpe_stream_t stream);                                  /* :127-171 */
/* transposes of the three builders above (autograd of the stand-alone modules' get_bias() / get_freqs_cis(), which
 * the reference returns as differentiable tensors): dtable [H,2L-1], dcoeff [deg+1] or [H,deg+1], dfreqs [2,H,half]
 * are OVERWRITTEN with the gradient w.r.t. the parameter given d bias [H,L,L] resp. d cos / d sin [H,P,half].       */
int vitpe_relative_bias_bwd(const float* dbias, float* dtable, int H, int L, vitpe_stream_t stream);
int vitpe_polynomial_bias_bwd(const float* dbias, float* dcoeff, int H, int G, int degree, int per_head,
                              vitpe_stream_t stream);
int vitpe_rope_mixed_tables_bwd(const float* freqs, const float* dcos, const float* dsin, float* dfreqs, int H,
                                int G, int half, vitpe_stream_t stream);
/* models/rope_utils.py:3-37 on x [B,H,P,HD] fp32 (called once for q, once for k)              */
int vitpe_apply_rotary(const float* x, float* y, const float* cosv, const float* sinv, int B, int H,
                       int P, int HD, int per_head, vitpe_stream_t stream);

/* ---- classifier head + loss (vit.py:284-285, train.py:113,119-121,194) -------------------- */
int vitpe_head_fwd(int dtype, const void* x, const float* gamma, const float* beta, const float* Wh,
                   const float* bh, float* logits, float* ws_xhat, float* ws_yn, float* ws_rstd,
                   int B, int Ntok, int D, int Cn, float eps, vitpe_stream_t stream);
/* out2[0] = mean CE, out2[1] = #correct; dlogits (NULL to skip) = (softmax-onehot)*grad_scale */
int vitpe_cross_entropy(const float* logits, const long long* labels, float* dlogits, float* out2,
                        int B, int Cn, float grad_scale, vitpe_stream_t stream);
/* The same with its scalars read ON THE DEVICE (a captured step follows a ragged last batch; reference train.py:89-90
 * has no drop_last): ctl = {grad_scale, loss_scale, n_valid}.  Rows b >= n_valid are padding: dlogits = 0 (so every
 * gradient downstream is untouched by them), left out of loss and accuracy.  out2[0] = loss_scale * sum of the valid
 * rows' losses, out2[1] = #correct; metric_acc (nullable) += out2.                                                  */
int vitpe_cross_entropy_ctl(const float* logits, const long long* labels, float* dlogits, float* out2,
                            float* metric_acc, const float* ctl, int B, int Cn, vitpe_stream_t stream);
int vitpe_head_bwd(int dtype, const float* dlogits, const float* Wh, const float* gamma,
                   const float* ws_xhat, const float* ws_yn, const float* ws_rstd, float* ws_dyn,
                   void* dx, float* dWh, float* dbh, float* dgamma, float* dbeta, int B, int Ntok,
                   int D, int Cn, vitpe_stream_t stream);
/* Training-step fusion of vitpe_head_fwd + vitpe_cross_entropy + vitpe_head_bwd (vit.py:284-285, train.py:113-114
 * and their autograd) for classes <= 64 (else hipErrorNotSupported): logits, dlogits = (softmax - onehot) *
 * grad_scale, dx (class row; other rows zero), out2 = [mean loss, #correct] of THIS batch, metric_acc (nullable)
 * += out2, parameter gradients accumulated.  scratch: 4 zero-initialised floats owned by the caller (batch
 * totals + arrival counter; the kernel re-arms it, so graph replay needs no memset).                      */
int vitpe_head_loss(int dtype, const void* x, const float* gamma, const float* beta, const float* Wh,
                    const float* bh, const long long* labels, float* logits, float* dlogits, float* ws_xhat,
                    float* ws_yn, float* ws_dyn, void* dx, float* out2, float* metric_acc, float* scratch,
                    float* dWh, float* dbh, float* dgamma, float* dbeta, int B, int Ntok, int D, int Cn,
                    float eps, float grad_scale, vitpe_stream_t stream);

/* The train step's head: vitpe_head_fwd + vitpe_cross_entropy_ctl + vitpe_head_bwd in one launch pair (classes <= 64,
 * D <= 768, else hipErrorNotSupported).  ctl as vitpe_cross_entropy_ctl.  dx: ONLY the class rows are written -- the
 * caller keeps rows 1.. of every image zero (they never change).  per_image: [B,2] work buffer ((loss, correct) per
 * image, summed in fixed order: no atomics on the totals).  Parameter gradients are accumulated.  hp_tick (nullable): the
 * optimizer's hp block -- the launch also advances its step counter and bias corrections ([5..7]), which lets the following
 * vitpe_adamw_step skip its own one-thread launch (zero_grad bit 1).                                              */
int vitpe_head_step(int dtype, const void* x, const float* gamma, const float* beta, const float* Wh,
                    const float* bh, const long long* labels, float* logits, float* dlogits, float* ws_xhat,
                    float* ws_yn, float* ws_dyn, void* dx, float* out2, float* metric_acc, float* per_image,
                    const float* ctl, float* dWh, float* dbh, float* dgamma, float* dbeta, int B, int Ntok,
                    int D, int Cn, float eps, float* hp_tick, vitpe_stream_t stream);

/* ---- optimizer + weight shadows (train.py:116,195) ------------------------------------------
 * hp (device, 16 floats): [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay [5]=step
 * [6]=bias_correction1 [7]=bias_correction2 [8]=grad_scale.  The step counter lives on the
 * device so the call can be replayed from a hipGraph.  zero_grad: bit 0 = clear g after the update,
 * bit 1 = the step counter / bias corrections were already advanced (vitpe_head_step hp_tick).   */
int vitpe_adamw_step(float* p, float* g, float* m, float* v, void* shadow_bf16, float* hp,
                     long long n, int zero_grad, vitpe_stream_t stream);
int vitpe_cast(int dtype, const float* src, void* dst, long long n, vitpe_stream_t stream);
int vitpe_transpose_cast(int dtype, const float* src, void* dst, int R, int C, vitpe_stream_t stream);
/* every weight shadow of the model in one launch.  desc: device array of ndesc records
 * {int64 src_off (elements into flat), int64 dst_off, int64 dst_off2 (elements into dst_base), int32 R, C, tile0,
 *  kind, HD, kind2, HD2, pad}: the fp32 matrix [R,C] at src_off is written as shadow `kind` at dst_off and, when
 *  kind2 >= 0, as shadow `kind2` at dst_off2 (one load of the source tile for both).  kind: 0 = transpose,
 *  1 = vitpe_pack_qkv_weights layout (HD = head dim), 2 / 3 = vitpe_pack_weight_frags of the matrix (natural / phi k
 *  order, HD = k chunk), 4 / 5 = vitpe_pack_weight_frags of its TRANSPOSE, 6 = vitpe_pack_qkv_weights_wide (HD = 32).  tile0 = running sum of
 *  ceil(R/32)*ceil(C/32); total_tiles = that sum over all records; tile_map (nullable): uint16[total_tiles], the record
 *  index of every 32x32 tile (without it each workgroup scans the records).                                        */
int vitpe_refresh_shadows(int dtype, const float* flat, void* dst_base, const void* desc, int ndesc,
                          int total_tiles, const unsigned short* tile_map, vitpe_stream_t stream);

/* Self-tests and residency / phase-census instrumentation are NOT part of this boundary: include/vitpe_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* VITPE_H */
