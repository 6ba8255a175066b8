// Fused MLP branch of a transformer block (reference models/vit.py:116-118,124: x + mlp(norm2(x)), timm
// Mlp = fc1 -> GELU -> fc2), d = 192, bf16:
//
//   forward :  xn = LayerNorm2(x) ; u = xn W1^T + b1 ; h = gelu(u) ; out = x + h W2^T + b2   (+ LN stats of out)
//   backward:  du = (dy W2) * gelu'(u) ; dxn = du W1 ; dx = dy + LayerNorm2'(dxn)  (+ dgamma, dbeta)
//              -- the same two-GEMM pipeline on the transposed weight shadows: chunk j of du is stored once
//              (the fc1 weight gradient reads it) and consumed from LDS; the residual is dy itself.
//
// One workgroup (12 waves, 3 x 4, wave tile 48 x 48) owns a panel of <= 144 token rows through BOTH
// GEMMs.  The hidden activation is produced and consumed in chunks of 192 columns: chunk j of u comes
// out of GEMM1's accumulators, goes through bias + GELU in the epilogue (u and h are stored once for
// the backward pass) and is parked in LDS as bf16 in the A-operand image; GEMM2 accumulates
// out += h_j W2[:, j]^T straight from there.  Compared with two panel-GEMM launches the hidden
// activation is never re-read from HBM (-51 MB per layer at batch 512), LayerNorm(x) is staged once
// instead of once per column tile, and the residual is the panel the workgroup already loaded.
//
// LDS (159.7 KB of 160): normalised x panel 54 KB (A of GEMM1, 3 K-slabs of [144][64]), h chunk 54 KB
// (A of GEMM2), weight slabs 2 x 24 KB (double buffer; [192][64], global loads two slabs ahead in
// registers).  All images are the 128-B-row XOR-swizzled layout of gemm.hip (16-B slot ^= (row>>1)&7).
// The epilogues park the accumulators (fp32) in the idle weight buffers and work on 8-column
// pieces with 16-B coalesced accesses, exactly like the panel GEMM.
#include "common.h"

namespace vitpe {

enum { MLP_FWD = 0, MLP_BWD = 1 };

// MLP_BWD reuses the fields: x = dy [M,192] (gradient of the block output), W1 = fc2.weight^T [HID,192],
// W2 = fc1.weight^T [192,HID], u_out = du [M,HID] (output), out = dx [M,192]; gamma/mean/rstd = norm2's
// weight and the statistics of its input rows ln_x; b1/b2/beta/xn_out/h_out/mean_out unused.
struct MlpFwdArgs {
  const void* x;        // [M,192] raw block input of the MLP branch (x_mid)
  const float* gamma;   // norm2 weight / bias [192]
  const float* beta;
  const float* mean;    // [M] LayerNorm statistics of x rows
  const float* rstd;
  void* xn_out;         // [M,192] LayerNorm(x) (for the weight gradient), nullable
  const void* W1;       // [HID,192] T
  const float* b1;      // [HID]
  const void* W2;       // [192,HID] T
  const float* b2;      // [192]
  void* u_out;          // [M,HID] pre-activation (backward)
  void* h_out;          // [M,HID] gelu(u) (fc2 weight gradient)
  void* out;            // [M,192] x + mlp(LN(x))
  float* mean_out;      // optional LayerNorm statistics of the OUTPUT rows (next block's norm1)
  float* rstd_out;
  int M, HID, panel_rows;
  float eps;
  // PRE (forward only): the attention branch's tail runs first in the same workgroup --
  //   x_mid = pre_r + pre_a Wp^T + pre_b (vit.py:91,122: proj + residual), stored to `xmid_out` together with its
  //   LayerNorm statistics (mean_w / rstd_w); `x`, `mean`, `rstd` are then not inputs.
  const void* pre_a;    // [M,192] merged-head attention output
  const void* pre_w;    // attn.proj.weight [192,192] T
  const float* pre_b;   // attn.proj.bias [192]
  const void* pre_r;    // [M,192] block input (residual)
  void* xmid_out;       // [M,192]
  float* mean_w;        // [M] statistics of x_mid rows (norm2), written
  float* rstd_w;
  float eps_pre;
  // POST (backward only): the data gradient of the attention projection runs last in the same workgroup --
  //   da = dx W_proj  (dx = this kernel's output rows), stored to `post_out` (input of the attention backward)
  const void* post_w;   // attn.proj.weight^T [192,192] T (transposed shadow)
  void* post_out;       // [M,192]
  const void* u_in;     // MLP_BWD: pre-activation u [M,HID] saved by the forward (or gelu'(u), see u_is_gprime)
  int u_is_gprime;      // MLP_BWD: u_in holds gelu'(u) (saved by vitpe_block_tail2_fwd): multiply, no erf here
  const void* ln_x;     // MLP_BWD: LayerNorm input rows (x_mid) [M,192]
  float* dgamma;        // MLP_BWD: accumulated (fp32 atomics, one per column and workgroup)
  float* dbeta;
};

constexpr int MLP_D = 192, MLP_BM = 144, MLP_ROWB = 128;

template <typename T, int MODE, bool PRE>   // PRE: forward = projection prologue, backward = projection-gradient epilogue
__global__ __launch_bounds__(768) void mlp_fwd_kernel(MlpFwdArgs a) {
  constexpr bool POST = PRE && MODE == MLP_BWD;
  constexpr bool FPRE = PRE && MODE == MLP_FWD;
  static_assert(sizeof(T) == 2, "bf16 only: the fp32 images would not fit LDS (the engine's fp32 mode runs unfused)");
  constexpr int D = MLP_D, BM = MLP_BM, BN = 192, ROWB = MLP_ROWB, CHN = 8;
  constexpr int SLAB_A = BM * ROWB;           // one K-slab of an A image: [144][64] bf16
  constexpr int SLAB_W = BN * ROWB;           // one weight slab: [192][64] bf16
  __shared__ __attribute__((aligned(16))) unsigned char sXA[3 * SLAB_A];
  __shared__ __attribute__((aligned(16))) unsigned char sHB[3 * SLAB_A];
  __shared__ __attribute__((aligned(16))) unsigned char sWB[2 * SLAB_W];

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int M = a.M, HID = a.HID;
  const int m0 = blockIdx.x * a.panel_rows, m_end = min(M, m0 + a.panel_rows);
  const T* __restrict__ X = reinterpret_cast<const T*>(FPRE ? a.xmid_out : a.x);   // forward PRE: x_mid is produced here
  const T* __restrict__ W1 = reinterpret_cast<const T*>(a.W1);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.W2);
  const Chunk16 zero = {0u, 0u, 0u, 0u};
  const int nchunk = HID / BN;                // hidden chunks
  constexpr int S0 = FPRE ? 3 : 0;            // forward PRE: three slabs of the projection weight first
  const int S1 = S0 + nchunk * 6;             // then per chunk 3 slabs of W1 and 3 of W2
  const int S = S1 + (POST ? 3 : 0);          // backward POST: three slabs of the transposed projection weight last

  // ---- weight slab s -> registers (2 x 16 B per thread) -------------------------------------------
  auto gload = [&](Chunk16* r, int s) {
    const int sm = s - S0;
    const int j = sm / 6, ph = (sm % 6) / 3, ks = (sm + 6) % 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 768 * i, row = q >> 3, cc = q & 7;
      const T* src;
      if (FPRE && s < S0) src = reinterpret_cast<const T*>(a.pre_w) + (size_t)row * D + s * 64 + cc * CHN;
      else if (POST && s >= S1) src = reinterpret_cast<const T*>(a.post_w) + (size_t)row * D + (s - S1) * 64 + cc * CHN;
      else src = (ph == 0) ? W1 + (size_t)(j * BN + row) * D + ks * 64 + cc * CHN
                           : W2 + (size_t)row * HID + j * BN + ks * 64 + cc * CHN;
      r[i] = *reinterpret_cast<const Chunk16*>(src);
    }
  };
  auto sstore = [&](const Chunk16* r, int buf) {
    unsigned char* base = sWB + buf * SLAB_W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 768 * i, row = q >> 3, cc = q & 7;
      *reinterpret_cast<Chunk16*>(base + row * ROWB + ((cc ^ ((row >> 1) & 7)) << 4)) = r[i];
    }
  };
  // acc[nt][mt] += W-slab rows (A operand) x activation rows (B operand), one 64-deep K slab
  auto compute = [&](const unsigned char* sA, int buf, f32x4 (&acc)[3][3]) {
    const unsigned char* sW = sWB + buf * SLAB_W;
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
      Frag<T> fw[3], fa[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        const int row = wn * 48 + 16 * nt + c;
        fw[nt] = ld_frag(reinterpret_cast<const T*>(sW + row * ROWB + (((4 * cs + g) ^ ((row >> 1) & 7)) << 4)));
      }
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const int row = wm * 48 + 16 * mt + c;
        fa[mt] = ld_frag(reinterpret_cast<const T*>(sA + row * ROWB + (((4 * cs + g) ^ ((row >> 1) & 7)) << 4)));
      }
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) mma(fw[nt], fa[mt], acc[nt][mt]);
    }
  };

  // the first two weight slabs fly under the x staging
  Chunk16 ra[2], rb[2];
  gload(ra, 0);
  gload(rb, 1);

  // ---- stage LayerNorm(x panel) as the A image of GEMM1 (and write it out for the weight gradient) ----
  {
    constexpr int TOTAL = BM * 24;            // 16-B chunks of the panel
    constexpr int ITERS = (TOTAL + 767) / 768;
    Chunk16 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int q = tid + 768 * it, row = q / 24, cc = q % 24;
      v[it] = zero;
      if (q < TOTAL && m0 + row < m_end)
        v[it] = *reinterpret_cast<const Chunk16*>((FPRE ? reinterpret_cast<const T*>(a.pre_a) : X) + (size_t)(m0 + row) * D + cc * CHN);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int q = tid + 768 * it, row = q / 24, cc = q % 24;
      if (q < TOTAL) {
        if (MODE == MLP_FWD && !FPRE && m0 + row < m_end) {
          const float mean = a.mean[m0 + row], rstd = a.rstd[m0 + row];
          float f[CHN];
          chunk_to_f32<T>(v[it], f);
#pragma unroll
          for (int h4 = 0; h4 < 2; ++h4) {
            const f32x4 gq = *reinterpret_cast<const f32x4*>(a.gamma + cc * CHN + 4 * h4);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(a.beta + cc * CHN + 4 * h4);
#pragma unroll
            for (int t = 0; t < 4; ++t) f[4 * h4 + t] = (f[4 * h4 + t] - mean) * rstd * gq[t] + bq[t];
          }
          v[it] = f32_to_chunk<T>(f);
          if (a.xn_out != nullptr)
            __builtin_nontemporal_store(v[it], reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.xn_out) + (size_t)(m0 + row) * D + cc * CHN));
        }
        const int slab = cc >> 3, slot = cc & 7;
        *reinterpret_cast<Chunk16*>(sXA + slab * SLAB_A + row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4)) = v[it];
      }
    }
  }

  f32x4 acc1[3][3], acc2[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  constexpr int EP_LD = BN + 4;
  static_assert(48 * EP_LD * 4 <= 2 * SLAB_W, "parked tile must fit the weight buffers");
  float* ep = reinterpret_cast<float*>(sWB);
  T* __restrict__ Uo = reinterpret_cast<T*>(a.u_out);
  T* __restrict__ Ho = reinterpret_cast<T*>(a.h_out);
  T* __restrict__ Out = reinterpret_cast<T*>(a.out);

  // one pipeline step: slab s (held in ra) -> LDS, slab s+2 -> registers, MFMAs of slab s
  auto step = [&](int s, const unsigned char* sA, f32x4 (&acc)[3][3]) {
    sstore(ra, s & 1);
    ra[0] = rb[0]; ra[1] = rb[1];
    if (s + 2 < S) gload(rb, s + 2);
    __syncthreads();
    compute(sA, s & 1, acc);
  };

  if (FPRE) {
    // ---- GEMM0: attention branch tail  x_mid = x_in + a Wp^T + bp ; row statistics ; LN2 -> A image of GEMM1 ----
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) step(ks, sXA + ks * SLAB_A, acc1);
    const float invN0 = 1.0f / (float)BN;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * 48 + 16 * nt + 4 * g) = acc1[nt][pass];
      __syncthreads();
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {   // 32 lanes per row (24 live)
        const int row = (tid >> 5) + 24 * i, pc = tid & 31;
        const bool live = pc < 24;
        const int lrow = (row >> 4) * 48 + 16 * pass + (row & 15);
        const int gm = m0 + lrow, gn = pc * 8;
        const bool ok = live && gm < m_end;
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = 0.f;
        if (ok) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.pre_b + gn);
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.pre_b + gn + 4);
          float rv[8];
          chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.pre_r) + (size_t)gm * D + gn), rv);
#pragma unroll
          for (int t = 0; t < 4; ++t) { v[t] = x0[t] + b0[t] + rv[t]; v[4 + t] = x1[t] + b1v[t] + rv[4 + t]; }
          const Chunk16 xc = f32_to_chunk<T>(v);
          *reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.xmid_out) + (size_t)gm * D + gn) = xc;
          chunk_to_f32<T>(xc, v);   // statistics and LayerNorm of the values as stored
        }
        float sres = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) sres += v[t];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sres += __shfl_xor(sres, o, 64);
        const float mean = sres * invN0;
        float sq = 0.f;
        if (ok) {
#pragma unroll
          for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; sq += d * d; }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = 1.0f / sqrtf(sq * invN0 + a.eps_pre);
        if (pc == 0 && gm < m_end) { a.mean_w[gm] = mean; a.rstd_w[gm] = rstd; }
        if (live) {
          Chunk16 xn = zero;
          if (ok) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + gn);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.gamma + gn + 4);
            const f32x4 be0 = *reinterpret_cast<const f32x4*>(a.beta + gn);
            const f32x4 be1 = *reinterpret_cast<const f32x4*>(a.beta + gn + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              v[t] = (v[t] - mean) * rstd * g0[t] + be0[t];
              v[4 + t] = (v[4 + t] - mean) * rstd * g1[t] + be1[t];
            }
            xn = f32_to_chunk<T>(v);
            if (a.xn_out != nullptr)
              __builtin_nontemporal_store(xn, reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.xn_out) + (size_t)gm * D + gn));
          }
          const int slab = pc >> 3, slot = pc & 7;
          *reinterpret_cast<Chunk16*>(sXA + slab * SLAB_A + lrow * ROWB + ((slot ^ ((lrow >> 1) & 7)) << 4)) = xn;
        }
      }
    }
    __syncthreads();   // parked tile consumed, A image of GEMM1 complete
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc1[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  for (int j = 0; j < nchunk; ++j) {
    // ---- GEMM1: u_j = LN(x) W1[j]^T -------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) step(S0 + 6 * j + ks, sXA + ks * SLAB_A, acc1);
    // ---- epilogue 1: + b1, store u, GELU, store h, park h (bf16) as GEMM2's A image ----------------
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * 48 + 16 * nt + 4 * g) = acc1[nt][pass];
      __syncthreads();
#pragma unroll 1   // rolled: both iterations' temporaries in flight would spill (two accumulator sets are live)
      for (int i = 0; i < 2; ++i) {
        const int qd = tid + 768 * i;
        if (qd < 48 * 24) {
          const int row = qd / 24, pc = qd % 24;
          const int lrow = (row >> 4) * 48 + 16 * pass + (row & 15);   // row inside the panel
          const int gm = m0 + lrow, gn = j * BN + pc * 8;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
          float v[8];
          Chunk16 hc;
          if (MODE == MLP_FWD) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b1 + gn);
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + gn + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] = x0[t] + b0[t]; v[4 + t] = x1[t] + b1v[t]; }
            const Chunk16 uc = f32_to_chunk<T>(v);
            gelu_erf_x8(v);
            hc = f32_to_chunk<T>(v);
            if (gm < m_end) {
              __builtin_nontemporal_store(uc, reinterpret_cast<Chunk16*>(Uo + (size_t)gm * HID + gn));
              *reinterpret_cast<Chunk16*>(Ho + (size_t)gm * HID + gn) = hc;
            }
          } else {   // du = (dy W2) * gelu'(u)
            float uv[8];
            chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.u_in) + (size_t)min(gm, m_end - 1) * HID + gn), uv);
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] = x0[t]; v[4 + t] = x1[t]; }
            if (a.u_is_gprime) {
#pragma unroll
              for (int t = 0; t < 8; ++t) v[t] *= uv[t];
            } else {
              gelu_erf_grad_mul_x8(v, uv);
            }
            hc = f32_to_chunk<T>(v);
            if (gm < m_end) *reinterpret_cast<Chunk16*>(Uo + (size_t)gm * HID + gn) = hc;
          }
          const int slab = pc >> 3, slot = pc & 7;
          *reinterpret_cast<Chunk16*>(sHB + slab * SLAB_A + lrow * ROWB + ((slot ^ ((lrow >> 1) & 7)) << 4)) = hc;
        }
      }
    }
    __syncthreads();   // parked tile consumed before the next weight slab overwrites it
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc1[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- GEMM2: out += h_j W2[:, j]^T  (the step's barrier also orders the h image writes) ----------
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) step(S0 + 6 * j + 3 + ks, sHB + ks * SLAB_A, acc2);
  }

  if (MODE == MLP_BWD) {
    // ---- epilogue 2 (backward): tile = dxn, the gradient of the LayerNorm output.  dx = dy + rstd*(g - mean(g)
    // - xhat*mean(g*xhat)), g = dxn*gamma, xhat = (x - mean)*rstd; dgamma += dxn*xhat, dbeta += dxn (column sums).
    const float invN = 1.0f / (float)BN;
    float lnacc_g[8], lnacc_b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { lnacc_g[t] = 0.f; lnacc_b[t] = 0.f; }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * 48 + 16 * nt + 4 * g) = acc2[nt][pass];
      __syncthreads();
#pragma unroll 1
      for (int i = 0; i < 2; ++i) {   // 32 lanes per row (24 live): the row means reduce with shuffles
        const int row = (tid >> 5) + 24 * i, pc = tid & 31;
        const bool live = pc < 24;
        const int gm = m0 + (row >> 4) * 48 + 16 * pass + (row & 15), gn = pc * 8;
        const bool ok = live && gm < m_end;
        const int gmc = min(gm, m_end - 1);
        const size_t off = (size_t)gmc * D + gn;
        const float mean = a.mean[gmc], rstd = a.rstd[gmc];
        float v[8], xh[8], gv[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { v[t] = 0.f; xh[t] = 0.f; gv[t] = 0.f; }
        if (live) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
          float xv[8];
          chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.ln_x) + off), xv);
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + gn);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.gamma + gn + 4);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            // the gradient as the unfused path sees it: dxn rounded to T
            v[t] = to_f32(from_f32<T>(t < 4 ? x0[t & 3] : x1[t & 3]));
            xh[t] = (xv[t] - mean) * rstd;
            gv[t] = v[t] * (t < 4 ? g0[t & 3] : g1[t & 3]);
            s1 += gv[t];
            s2 += gv[t] * xh[t];
          }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 *= invN;
        s2 *= invN;
        if (ok) {
          float rv[8], o8[8];
          chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(X + off), rv);   // residual gradient = dy itself
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            o8[t] = rstd * (gv[t] - s1 - xh[t] * s2) + rv[t];
            lnacc_g[t] += v[t] * xh[t];
            lnacc_b[t] += v[t];
          }
          const Chunk16 oc = f32_to_chunk<T>(o8);
          *reinterpret_cast<Chunk16*>(Out + off) = oc;
          if (POST) {   // dx rows as the A image of the projection-gradient GEMM (the x panel image is free by now)
            const int lrow = gm - m0, slab = pc >> 3, slot = pc & 7;
            *reinterpret_cast<Chunk16*>(sXA + slab * SLAB_A + lrow * ROWB + ((slot ^ ((lrow >> 1) & 7)) << 4)) = oc;
          }
        }
      }
    }
    // column sums: thread (row group tid>>5, piece tid&31) -> LDS [24][2][192] -> one atomic per column
    __syncthreads();
    float* red = reinterpret_cast<float*>(sWB);
    {
      const int pc = tid & 31, grp = tid >> 5;
      if (pc < 24) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          red[(grp * 2 + 0) * BN + pc * 8 + t] = lnacc_g[t];
          red[(grp * 2 + 1) * BN + pc * 8 + t] = lnacc_b[t];
        }
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, col = tid % BN;
      float sres = 0.f;
#pragma unroll
      for (int k = 0; k < 24; ++k) sres += red[(k * 2 + which) * BN + col];
      atomicAdd((which == 0 ? a.dgamma : a.dbeta) + col, sres);
    }
    if (POST) {
      // ---- GEMM3: da = dx W_proj, plain store (rows >= m_end of the image hold stale finite values: never stored)
      __syncthreads();   // column-sum scratch (weight buffers) consumed
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc1[i][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) step(S1 + ks, sXA + ks * SLAB_A, acc1);
      T* __restrict__ Po = reinterpret_cast<T*>(a.post_out);
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * 48 + 16 * nt + 4 * g) = acc1[nt][pass];
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          const int qd = tid + 768 * i;
          if (qd < 48 * 24) {
            const int row = qd / 24, pc = qd % 24;
            const int gm = m0 + (row >> 4) * 48 + 16 * pass + (row & 15);
            if (gm < m_end) {
              float v[8];
              const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
              const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
#pragma unroll
              for (int t = 0; t < 4; ++t) { v[t] = x0[t]; v[4 + t] = x1[t]; }
              *reinterpret_cast<Chunk16*>(Po + (size_t)gm * D + pc * 8) = f32_to_chunk<T>(v);
            }
          }
        }
      }
    }
    return;
  }

  // ---- epilogue 2: + b2 + residual (the raw x rows), store, LayerNorm statistics of the output ----
  const float invN = 1.0f / (float)BN;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * 48 + 16 * nt + 4 * g) = acc2[nt][pass];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // 32 lanes per row (24 live): row statistics reduce with shuffles
      const int row = (tid >> 5) + 24 * i, pc = tid & 31;
      const bool live = pc < 24;
      const int gm = m0 + (row >> 4) * 48 + 16 * pass + (row & 15), gn = pc * 8;
      const bool ok = live && gm < m_end;
      float v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = 0.f;
      if (ok) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b2 + gn);
        const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b2 + gn + 4);
        float rv[8];
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(X + (size_t)gm * D + gn), rv);
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = x0[t] + b0[t] + rv[t]; v[4 + t] = x1[t] + b1v[t] + rv[4 + t]; }
        const Chunk16 oc = f32_to_chunk<T>(v);
        *reinterpret_cast<Chunk16*>(Out + (size_t)gm * D + gn) = oc;
        chunk_to_f32<T>(oc, v);   // statistics of the values as stored
      }
      if (a.mean_out != nullptr) {
        float sres = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) sres += v[t];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sres += __shfl_xor(sres, o, 64);
        const float mean = sres * invN;
        float sq = 0.f;
        if (ok) {
#pragma unroll
          for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; sq += d * d; }
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
        if (pc == 0 && gm < m_end) {
          a.mean_out[gm] = mean;
          a.rstd_out[gm] = 1.0f / sqrtf(sq * invN + a.eps);
        }
      }
    }
  }
}

}  // namespace vitpe

using namespace vitpe;

static int mlp_launch(int mode, MlpFwdArgs& a, hipStream_t stream, bool pre = false) {
  // as many panels as CUs (x waves of them), each <= 144 rows
  const int waves = (a.M + 256 * MLP_BM - 1) / (256 * MLP_BM);
  const int npanels = 256 * waves;
  int rows = (a.M + npanels - 1) / npanels;
  a.panel_rows = rows < 16 ? 16 : rows;
  const int grid = (a.M + a.panel_rows - 1) / a.panel_rows;
  if (mode == MLP_FWD && pre) hipLaunchKernelGGL((mlp_fwd_kernel<bf16, MLP_FWD, true>), dim3(grid), dim3(768), 0, stream, a);
  else if (mode == MLP_FWD) hipLaunchKernelGGL((mlp_fwd_kernel<bf16, MLP_FWD, false>), dim3(grid), dim3(768), 0, stream, a);
  else if (pre) hipLaunchKernelGGL((mlp_fwd_kernel<bf16, MLP_BWD, true>), dim3(grid), dim3(768), 0, stream, a);
  else hipLaunchKernelGGL((mlp_fwd_kernel<bf16, MLP_BWD, false>), dim3(grid), dim3(768), 0, stream, a);
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_mlp_fwd_supported(int dtype, int D, int HID) {
  return dtype == 1 && D == MLP_D && HID > 0 && HID % 192 == 0;
}

extern "C" int vitpe_mlp_fwd(int dtype, const void* x, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, void* xn_out, const void* W1, const float* b1, const void* W2,
                             const float* b2, void* u_out, void* h_out, void* out, float* mean_out, float* rstd_out,
                             float eps, int M, int D, int HID, hipStream_t stream) {
  VITPE_REQUIRE(x && gamma && beta && mean && rstd && W1 && b1 && W2 && b2 && u_out && h_out && out && M >= 0);
  VITPE_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr));
  if (!vitpe_mlp_fwd_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  MlpFwdArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.xn_out = xn_out; a.W1 = W1; a.b1 = b1;
  a.W2 = W2; a.b2 = b2; a.u_out = u_out; a.h_out = h_out; a.out = out; a.mean_out = mean_out; a.rstd_out = rstd_out;
  a.M = M; a.HID = HID; a.eps = eps;
  return mlp_launch(MLP_FWD, a, stream);
}

// Backward of the MLP branch w.r.t. its input (the weight gradients are vitpe_wgrad_group problems on du / dy):
//   du = (dy fc2.weight) * gelu'(u)  [M,HID] (stored) ; dx = dy + LayerNorm'(du fc1.weight) ; dgamma/dbeta accumulated.
// W2t = fc2.weight^T [HID,192], W1t = fc1.weight^T [192,HID] (transposed shadows), x = LayerNorm input rows with
// statistics mean/rstd.  Same support set as vitpe_mlp_fwd.
extern "C" int vitpe_mlp_bwd(int dtype, const void* dy, const void* u, const void* W2t, const void* W1t, const void* x,
                             const float* mean, const float* rstd, const float* gamma, void* du, void* dx,
                             float* dgamma, float* dbeta, int M, int D, int HID, hipStream_t stream) {
  VITPE_REQUIRE(dy && u && W2t && W1t && x && mean && rstd && gamma && du && dx && dgamma && dbeta && M >= 0);
  if (!vitpe_mlp_fwd_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  MlpFwdArgs a{};
  a.x = dy; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.W1 = W2t; a.W2 = W1t; a.u_out = du; a.out = dx;
  a.u_in = u; a.ln_x = x; a.dgamma = dgamma; a.dbeta = dbeta; a.M = M; a.HID = HID;
  return mlp_launch(MLP_BWD, a, stream);
}

// Attention-branch tail + MLP branch of a block in one kernel (vit.py:91,122-124):
//   x_mid = x_in + attn_out Wp^T + bp ; out = x_mid + fc2(gelu(fc1(LayerNorm2(x_mid))))
// x_mid and its LayerNorm statistics (mean2/rstd2) are outputs (the backward pass needs them); the rest as vitpe_mlp_fwd.
extern "C" int vitpe_block_tail_fwd(int dtype, const void* attn_out, const void* x_in, const void* Wp, const float* bp,
                                    const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                                    void* xn_out, const void* W1, const float* b1, const void* W2, const float* b2,
                                    void* u_out, void* h_out, void* out, float* mean_out, float* rstd_out, float eps2,
                                    float eps_next, int M, int D, int HID, hipStream_t stream) {
  VITPE_REQUIRE(attn_out && x_in && Wp && bp && gamma && beta && x_mid && mean2 && rstd2 && W1 && b1 && W2 && b2 &&
                u_out && h_out && out && M >= 0);
  VITPE_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr));
  if (!vitpe_mlp_fwd_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  MlpFwdArgs a{};
  a.pre_a = attn_out; a.pre_r = x_in; a.pre_w = Wp; a.pre_b = bp; a.xmid_out = x_mid; a.mean_w = mean2; a.rstd_w = rstd2;
  a.eps_pre = eps2; a.gamma = gamma; a.beta = beta; a.xn_out = xn_out; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2;
  a.u_out = u_out; a.h_out = h_out; a.out = out; a.mean_out = mean_out; a.rstd_out = rstd_out;
  a.M = M; a.HID = HID; a.eps = eps_next;
  return mlp_launch(MLP_FWD, a, stream, true);
}

// Backward mirror of vitpe_block_tail_fwd's fusion: vitpe_mlp_bwd plus the data gradient of the attention projection,
//   da = dx W_proj   [M,192]  (WpT = attn.proj.weight^T, the transposed shadow), the input of the attention backward.
static int block_tail_bwd_impl(int u_is_gprime, int dtype, const void* dy, const void* u, const void* W2t, const void* W1t,
                                    const void* x, const float* mean, const float* rstd, const float* gamma, void* du,
                                    void* dx, float* dgamma, float* dbeta, const void* WpT, void* da, int M, int D,
                                    int HID, hipStream_t stream) {
  VITPE_REQUIRE(dy && u && W2t && W1t && x && mean && rstd && gamma && du && dx && dgamma && dbeta && WpT && da && M >= 0);
  if (!vitpe_mlp_fwd_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  MlpFwdArgs a{};
  a.x = dy; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.W1 = W2t; a.W2 = W1t; a.u_out = du; a.out = dx;
  a.u_in = u; a.ln_x = x; a.dgamma = dgamma; a.dbeta = dbeta; a.post_w = WpT; a.post_out = da; a.M = M; a.HID = HID;
  a.u_is_gprime = u_is_gprime;
  return mlp_launch(MLP_BWD, a, stream, true);
}
extern "C" int vitpe_block_tail_bwd(int dtype, const void* dy, const void* u, const void* W2t, const void* W1t,
                                    const void* x, const float* mean, const float* rstd, const float* gamma, void* du,
                                    void* dx, float* dgamma, float* dbeta, const void* WpT, void* da, int M, int D,
                                    int HID, hipStream_t stream) {
  return block_tail_bwd_impl(0, dtype, dy, u, W2t, W1t, x, mean, rstd, gamma, du, dx, dgamma, dbeta, WpT, da, M, D, HID, stream);
}
// The same with gp = gelu'(u) [M,HID] in place of u (what vitpe_block_tail2_fwd saves): du = (dy W2) * gp, no erf.
extern "C" int vitpe_block_tail_bwd_gp(int dtype, const void* dy, const void* gp, const void* W2t, const void* W1t,
                                       const void* x, const float* mean, const float* rstd, const float* gamma, void* du,
                                       void* dx, float* dgamma, float* dbeta, const void* WpT, void* da, int M, int D,
                                       int HID, hipStream_t stream) {
  return block_tail_bwd_impl(1, dtype, dy, gp, W2t, W1t, x, mean, rstd, gamma, du, dx, dgamma, dbeta, WpT, da, M, D, HID, stream);
}
