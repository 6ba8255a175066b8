// Fused attention-with-positional-encoding, forward and backward (the north-star kernels).
//
// Forward replaces, per layer, reference models/vit.py:47-88:
//   qkv Linear (no bias) -> head split -> RoPE rotate-half on patch tokens (rope_utils.py:18-37,
//   class token skipped, vit.py:56-68) -> QK^T * hd^-0.5 (vit.py:71,75) -> + relative / polynomial
//   bias (vit.py:78-81; positional_encoding.py:82-95 / 127-171 evaluated on the fly from the
//   [H,2N-1] table / the coefficients: the [H,N,N] bias is never materialised) -> softmax ->
//   @V -> merged-head [B,N,d].
// Backward is the hand-derived autograd of the same (the reference has no explicit backward):
//   recompute q,k,v and the probabilities, dV, dP, dS, dQ, dK, inverse rotation, and the
//   positional-parameter gradients (relative table scatter-add, polynomial coefficients,
//   RoPE-mixed frequencies through cos/sin and the reference's view-scramble).
//
// Mapping: one workgroup (6 wavefronts) per image.  The layer-normed token matrix x[N,d] is loaded
// once with coalesced 16-B reads into LDS; heads are processed HPP at a time: three waves per head
// project q / k / v with MFMA (weight fragments straight from L2 into registers, x fragments from
// LDS), rotate in registers (the rotate-half partner j+hd/2 sits in the same lane of the
// neighbouring accumulator tile), and park q,k,v in LDS.  The attention core then runs one
// (head, 16-query tile) job per wave with the *swapped* product S^T = K Q^T so that each lane
// owns one query column: softmax row-max / row-sum are in-lane reductions plus two wavefront
// shuffles, and the probabilities feed the P.V MFMA directly from registers (V is read
// column-wise with ds_read_b64_tr_b16).
#include "common.h"

namespace vitpe {

struct AttnArgs {
  const void* xn;      // [B,N,D] T, layer-normed tokens
  const void* wqkv;    // [3D,D] T
  void* out;           // fwd: [B,N,D] T merged heads ; bwd: d_qkv [B,N,3D] T
  const void* dout;    // bwd: [B,N,D] T gradient of the merged-head output
  const float* cos;    // rope: axial [P,HD/2], mixed [H,P,HD/2] (contiguous)
  const float* sin;
  const float* table;  // relative: [H,2N-1]
  const float* coeff;  // polynomial: [deg+1] or [H,deg+1]
  float* dtable;       // bwd relative: [H,2N-1] accumulated (atomics)
  float* dcoeff;       // bwd polynomial: same shape as coeff, accumulated
  float* dfreqs;       // bwd rope-mixed: [2,H,HD/2] accumulated
  int B, N;
  int mode, grid, degree, coeff_per_head;
  float scale;
};

template <typename T, int HD, int D, int MT, int HPP>
struct AttnCfg {
  static constexpr int H = D / HD;
  static constexpr int NT = HD / 16;              // 16-wide feature tiles per head
  static constexpr int KS = D / 32;               // K32 chunks of the projection
  static constexpr int HC = HD / 32;              // K32 chunks over the head dim
  static constexpr int NP = 16 * MT;              // padded tokens
  static constexpr int SC = (MT + 1) / 2;         // K32 chunks over tokens
  static constexpr int VR = 32 * SC;              // rows incl. the zero tail read by 32-deep token contractions
  static constexpr int LDX = D + Pad<T>::elems;
  static constexpr int LDH = HD + Pad<T>::elems;
  static constexpr int HSZ = VR * LDH;            // one (matrix, head) LDS tile with the zero tail
  static constexpr int QSZ = NP * LDH;            // same without the tail (row-read operands only)
  static constexpr int TABLD = 2 * NP;
  static constexpr int MAXDEG = 7;
  static_assert(HD % 32 == 0 && D % HD == 0 && D % 32 == 0, "shape");
};

// additive logit bias for (query i, key j) of head h; 0 for rope / none / absolute
template <typename C>
VITPE_DEV float pe_bias(const AttnArgs& a, const float* s_tab, const float* s_coef, int h, int i, int j) {
  if (a.mode == PE_RELATIVE) {
    int idx = i - j + a.N - 1;  // positional_encoding.py:67-73 (1-D index incl. class token)
    idx = max(0, min(idx, 2 * a.N - 2));
    return s_tab[h * C::TABLD + idx];
  }
  if (a.mode == PE_POLY) {
    if (i < 1 || j < 1) return 0.f;  // class row / column stay zero (positional_encoding.py:165-169)
    const int pi = i - 1, pj = j - 1, G = a.grid;
    const int l1 = abs(pi % G - pj % G) + abs(pi / G - pj / G);
    const float* cf = s_coef + (a.coeff_per_head ? h * (C::MAXDEG + 1) : 0);
    const float x = (float)l1;
    float v = cf[a.degree];
    for (int k = a.degree - 1; k >= 0; --k) v = v * x + cf[k];
    return v;
  }
  return 0.f;
}

// ---- stage the image's tokens, PE tables; zero the tails --------------------------------
template <typename T, typename C>
VITPE_DEV void stage_tokens(const AttnArgs& a, int b, T* xs, T* hbuf, int hbuf_elems, float* s_tab, float* s_coef,
                            int nthreads) {
  constexpr int CHN = CH<T>::n;
  constexpr int DCH = C::LDX / CHN;  // chunks per LDS row incl. pad
  const int tid = threadIdx.x;
  const T* xg = reinterpret_cast<const T*>(a.xn) + (size_t)b * a.N * (C::LDX - Pad<T>::elems);
  const Chunk16 zero = {0u, 0u, 0u, 0u};
  constexpr int D = C::LDX - Pad<T>::elems;
  for (int q = tid; q < C::NP * DCH; q += nthreads) {
    const int row = q / DCH, cc = q % DCH;
    Chunk16 v = zero;
    if (row < a.N && cc * CHN < D) v = *reinterpret_cast<const Chunk16*>(xg + (size_t)row * D + cc * CHN);
    *reinterpret_cast<Chunk16*>(xs + row * C::LDX + cc * CHN) = v;
  }
  // zero every head tile once: rows >= NP are never written again and must read as 0
  for (int q = tid; q < hbuf_elems / CHN; q += nthreads) *reinterpret_cast<Chunk16*>(hbuf + q * CHN) = zero;
  if (a.mode == PE_RELATIVE) {
    for (int q = tid; q < C::H * C::TABLD; q += nthreads) {
      const int h = q / C::TABLD, i = q % C::TABLD;
      s_tab[q] = (i < 2 * a.N - 1) ? a.table[h * (2 * a.N - 1) + i] : 0.f;
    }
  }
  if (a.mode == PE_POLY) {
    for (int q = tid; q < C::H * (C::MAXDEG + 1); q += nthreads) {
      const int h = q / (C::MAXDEG + 1), k = q % (C::MAXDEG + 1);
      float v = 0.f;
      if (k <= a.degree) v = a.coeff_per_head ? a.coeff[h * (a.degree + 1) + k] : a.coeff[k];
      s_coef[q] = v;
    }
  }
}

// ---- QKV projection of one (head, matrix) by one wave, RoPE + scale, result to LDS --------
// Swapped MFMA orientation: A-operand = weight rows (feature n on the lane), B-operand = token rows,
// so acc[nt][tt][r] = qkv[token 16tt+c][feature 16nt+4g+r] and the rotate-half partner of
// feature f < HD/2 is acc[nt + NT/2] in the same lane and register.
template <typename T, typename C>
VITPE_DEV void project_head(const AttnArgs& a, const T* xs, T* dst, int h, int mat, int lane) {
  constexpr int D = C::LDX - Pad<T>::elems, HD = C::LDH - Pad<T>::elems;
  const int c = lane & 15, g = lane >> 4;
  const T* W = reinterpret_cast<const T*>(a.wqkv) + (size_t)(mat * D + h * HD) * D;
  f32x4 acc[C::NT][C::NP / 16];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
    for (int tt = 0; tt < C::NP / 16; ++tt) acc[nt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    Frag<T> wA[C::NT];
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) wA[nt] = ld_frag(W + (size_t)(16 * nt + c) * D + 32 * ks + 8 * g);
#pragma unroll
    for (int tt = 0; tt < C::NP / 16; ++tt) {
      const Frag<T> fb = ld_frag(xs + (16 * tt + c) * C::LDX + 32 * ks + 8 * g);
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) mma(wA[nt], fb, acc[nt][tt]);
    }
  }
  const bool rope = (a.mode == PE_ROPE_AXIAL || a.mode == PE_ROPE_MIXED) && mat < 2;
  if (rope) {
    const int P = a.N - 1;
    const size_t hoff = (a.mode == PE_ROPE_MIXED) ? (size_t)h * P * (HD / 2) : 0;
#pragma unroll
    for (int tt = 0; tt < C::NP / 16; ++tt) {
      const int tok = 16 * tt + c;
      if (tok >= 1 && tok < a.N) {  // class token is never rotated (vit.py:56-57)
#pragma unroll
        for (int nt = 0; nt < C::NT / 2; ++nt) {
          const size_t o = hoff + (size_t)(tok - 1) * (HD / 2) + 16 * nt + 4 * g;
          const f32x4 cs = *reinterpret_cast<const f32x4*>(a.cos + o);
          const f32x4 sn = *reinterpret_cast<const f32x4*>(a.sin + o);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x1 = acc[nt][tt][r], x2 = acc[nt + C::NT / 2][tt][r];
            acc[nt][tt][r] = x1 * cs[r] - x2 * sn[r];
            acc[nt + C::NT / 2][tt][r] = x1 * sn[r] + x2 * cs[r];
          }
        }
      }
    }
  }
  const float sc = (mat == 0) ? a.scale : 1.0f;
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
    for (int tt = 0; tt < C::NP / 16; ++tt)
      st4(dst + (16 * tt + c) * C::LDH + 16 * nt + 4 * g, acc[nt][tt][0] * sc, acc[nt][tt][1] * sc,
          acc[nt][tt][2] * sc, acc[nt][tt][3] * sc);
}

// =========================================================================================
// Forward
// =========================================================================================
template <typename T, int HD, int D, int MT, int HPP>
__global__ __launch_bounds__(384) void attn_fwd_kernel(AttnArgs a) {
  using C = AttnCfg<T, HD, D, MT, HPP>;
  __shared__ __attribute__((aligned(16))) T xs[C::NP * C::LDX];
  // q,k: [hh][NP][LDH] (row reads only) ; v: [hh][VR][LDH] (column reads run into the zero tail)
  constexpr int HB_ELEMS = HPP * (2 * C::QSZ + C::HSZ);
  __shared__ __attribute__((aligned(16))) T hb[HB_ELEMS];
  __shared__ __attribute__((aligned(16))) float s_tab[C::H * C::TABLD];
  __shared__ float s_coef[C::H * (C::MAXDEG + 1)];
  T* const qb = hb;
  T* const kb = hb + HPP * C::QSZ;
  T* const vb = hb + 2 * HPP * C::QSZ;

  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  stage_tokens<T, C>(a, b, xs, hb, HB_ELEMS, s_tab, s_coef, 384);
  __syncthreads();

  T* outp = reinterpret_cast<T*>(a.out);
  for (int h0 = 0; h0 < C::H; h0 += HPP) {
    {
      const int hh = wave / 3, mat = wave % 3, h = h0 + hh;
      T* dst = (mat == 0) ? qb + hh * C::QSZ : (mat == 1) ? kb + hh * C::QSZ : vb + hh * C::HSZ;
      if (hh < HPP && h < C::H) project_head<T, C>(a, xs, dst, h, mat, lane);
    }
    __syncthreads();
    for (int job = wave; job < HPP * MT; job += 6) {
      const int hh = job / MT, it = job % MT, h = h0 + hh;
      if (h >= C::H) continue;
      const T* qh = qb + hh * C::QSZ;
      const T* kh = kb + hh * C::QSZ;
      const T* vh = vb + hh * C::HSZ;
      Frag<T> bq[C::HC];
#pragma unroll
      for (int cs = 0; cs < C::HC; ++cs) bq[cs] = ld_frag(qh + (16 * it + c) * C::LDH + 32 * cs + 8 * g);
      f32x4 s[MT];
      const int i = 16 * it + c;
      float m = -1e30f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cs = 0; cs < C::HC; ++cs)
          mma(ld_frag(kh + (16 * jt + c) * C::LDH + 32 * cs + 8 * g), bq[cs], s[jt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + 4 * g + r;
          float v = s[jt][r] + pe_bias<C>(a, s_tab, s_coef, h, i, j);
          v = (j < a.N) ? v : -1e30f;
          s[jt][r] = v;
          m = fmaxf(m, v);
        }
      }
      m = xgroup_max(m);
      float l = 0.f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = expf(s[jt][r] - m);
          s[jt][r] = p;
          l += p;
        }
      l = xgroup_sum(l);
      f32x4 o[C::NT];
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sc = 0; sc < C::SC; ++sc) {
        const Frag<T> bp = acc_to_frag<T>(s[2 * sc], (2 * sc + 1 < MT) ? s[(2 * sc + 1 < MT) ? 2 * sc + 1 : 0] : z4);
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          mma(ld_frag_tr(vh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bp, o[dt]);
      }
      const float inv = 1.0f / l;
      if (i < a.N) {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          st4(outp + ((size_t)b * a.N + i) * D + h * HD + 16 * dt + 4 * g, o[dt][0] * inv, o[dt][1] * inv,
              o[dt][2] * inv, o[dt][3] * inv);
      }
    }
    __syncthreads();
  }
}

// =========================================================================================
// Backward
// =========================================================================================
template <typename T, int HD, int D, int MT, int HPP>
__global__ __launch_bounds__(384) void attn_bwd_kernel(AttnArgs a) {
  using C = AttnCfg<T, HD, D, MT, HPP>;
  __shared__ __attribute__((aligned(16))) T xs[C::NP * C::LDX];
  __shared__ __attribute__((aligned(16))) T hb[4 * HPP * C::HSZ];  // q,k,v,dO: [mat][hh][VR][LDH]
  __shared__ __attribute__((aligned(16))) float s_tab[C::H * C::TABLD];
  __shared__ float s_coef[C::H * (C::MAXDEG + 1)];
  __shared__ float s_stat[HPP * 2 * C::NP];        // [hh][lse | delta][token]
  __shared__ float s_dtab[C::H * C::TABLD];        // relative-table gradient of this image
  __shared__ float s_dcoef[C::H * (C::MAXDEG + 1)];
  __shared__ float s_dfreq[2 * C::H * (HD / 2)];

  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int P = a.N - 1;
  stage_tokens<T, C>(a, b, xs, hb, 4 * HPP * C::HSZ, s_tab, s_coef, 384);
  for (int q = threadIdx.x; q < C::H * C::TABLD; q += 384) s_dtab[q] = 0.f;
  for (int q = threadIdx.x; q < C::H * (C::MAXDEG + 1); q += 384) s_dcoef[q] = 0.f;
  for (int q = threadIdx.x; q < 2 * C::H * (HD / 2); q += 384) s_dfreq[q] = 0.f;
  __syncthreads();

  constexpr int CHN = CH<T>::n;
  const T* dog = reinterpret_cast<const T*>(a.dout) + (size_t)b * a.N * D;
  T* dq = reinterpret_cast<T*>(a.out) + (size_t)b * a.N * 3 * D;
  const bool rope = (a.mode == PE_ROPE_AXIAL || a.mode == PE_ROPE_MIXED);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (int h0 = 0; h0 < C::H; h0 += HPP) {
    {
      const int hh = wave / 3, mat = wave % 3, h = h0 + hh;
      if (hh < HPP && h < C::H) project_head<T, C>(a, xs, hb + (mat * HPP + hh) * C::HSZ, h, mat, lane);
    }
    // stage dO of the pass's heads (rows >= N stay zero from the initial clear)
    for (int q = threadIdx.x; q < HPP * a.N * (HD / CHN); q += 384) {
      const int hh = q / (a.N * (HD / CHN)), rem = q % (a.N * (HD / CHN));
      const int row = rem / (HD / CHN), cc = rem % (HD / CHN);
      if (h0 + hh < C::H)
        *reinterpret_cast<Chunk16*>(hb + (3 * HPP + hh) * C::HSZ + row * C::LDH + cc * CHN) =
            *reinterpret_cast<const Chunk16*>(dog + (size_t)row * D + (h0 + hh) * HD + cc * CHN);
    }
    __syncthreads();

    // ---- step 1: (head, query tile) jobs on the swapped tiles: stats, dS^T, dQ ------------
    for (int job = wave; job < HPP * MT; job += 6) {
      const int hh = job / MT, it = job % MT, h = h0 + hh;
      if (h >= C::H) continue;
      const T* qh = hb + (0 * HPP + hh) * C::HSZ;
      const T* kh = hb + (1 * HPP + hh) * C::HSZ;
      const T* vh = hb + (2 * HPP + hh) * C::HSZ;
      const T* doh = hb + (3 * HPP + hh) * C::HSZ;
      Frag<T> bq[C::HC], bdo[C::HC];
#pragma unroll
      for (int cs = 0; cs < C::HC; ++cs) {
        bq[cs] = ld_frag(qh + (16 * it + c) * C::LDH + 32 * cs + 8 * g);
        bdo[cs] = ld_frag(doh + (16 * it + c) * C::LDH + 32 * cs + 8 * g);
      }
      f32x4 s[MT], dp[MT];
      const int i = 16 * it + c;
      float m = -1e30f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        s[jt] = z4;
        dp[jt] = z4;
#pragma unroll
        for (int cs = 0; cs < C::HC; ++cs) {
          mma(ld_frag(kh + (16 * jt + c) * C::LDH + 32 * cs + 8 * g), bq[cs], s[jt]);
          mma(ld_frag(vh + (16 * jt + c) * C::LDH + 32 * cs + 8 * g), bdo[cs], dp[jt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + 4 * g + r;
          float v = s[jt][r] + pe_bias<C>(a, s_tab, s_coef, h, i, j);
          v = (j < a.N) ? v : -1e30f;
          s[jt][r] = v;
          m = fmaxf(m, v);
        }
      }
      m = xgroup_max(m);
      float l = 0.f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = expf(s[jt][r] - m);
          s[jt][r] = p;
          l += p;
        }
      l = xgroup_sum(l);
      const float inv = 1.0f / l;
      float dl = 0.f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[jt][r] *= inv;
          dl += s[jt][r] * dp[jt][r];
        }
      dl = xgroup_sum(dl);
      if (g == 0) {
        s_stat[(hh * 2 + 0) * C::NP + i] = m + logf(l);
        s_stat[(hh * 2 + 1) * C::NP + i] = dl;
      }
      // dS^T (in place of dp) and the bias-parameter gradients
      float cacc[C::MAXDEG + 1];
#pragma unroll
      for (int k = 0; k <= C::MAXDEG; ++k) cacc[k] = 0.f;
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + 4 * g + r;
          const float ds = (i < a.N && j < a.N) ? s[jt][r] * (dp[jt][r] - dl) : 0.f;
          dp[jt][r] = ds;
          if (a.mode == PE_RELATIVE) {
            if (i < a.N && j < a.N) atomicAdd(&s_dtab[h * C::TABLD + (i - j + a.N - 1)], ds);
          } else if (a.mode == PE_POLY) {
            if (i >= 1 && j >= 1 && i < a.N && j < a.N) {
              const int pi = i - 1, pj = j - 1, G = a.grid;
              const float x = (float)(abs(pi % G - pj % G) + abs(pi / G - pj / G));
              float pw = 1.f;
#pragma unroll
              for (int k = 0; k <= C::MAXDEG; ++k) {
                if (k <= a.degree) cacc[k] += ds * pw;
                pw *= x;
              }
            }
          }
        }
      if (a.mode == PE_POLY) {
#pragma unroll
        for (int k = 0; k <= C::MAXDEG; ++k) {
          if (k <= a.degree) {  // wave-uniform
            const float t = wave_sum(cacc[k]);
            if (lane == 0) atomicAdd(&s_dcoef[(a.coeff_per_head ? h : 0) * (C::MAXDEG + 1) + k], t);
          }
        }
      }
      // dQ~^T[d][i] = sum_j K^T[d][j] dS^T[j][i]
      f32x4 dqa[C::NT];
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt) dqa[dt] = z4;
#pragma unroll
      for (int sc = 0; sc < C::SC; ++sc) {
        const Frag<T> bs = acc_to_frag<T>(dp[2 * sc], (2 * sc + 1 < MT) ? dp[(2 * sc + 1 < MT) ? 2 * sc + 1 : 0] : z4);
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          mma(ld_frag_tr(kh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bs, dqa[dt]);
      }
      // inverse rotation (+ RoPE-mixed phase gradient), undo the folded scale, store the q part
      if (rope && i >= 1 && i < a.N) {
        const size_t hoff = (a.mode == PE_ROPE_MIXED) ? (size_t)h * P * (HD / 2) : 0;
#pragma unroll
        for (int nt = 0; nt < C::NT / 2; ++nt) {
          const size_t o = hoff + (size_t)(i - 1) * (HD / 2) + 16 * nt + 4 * g;
          const f32x4 cs = *reinterpret_cast<const f32x4*>(a.cos + o);
          const f32x4 sn = *reinterpret_cast<const f32x4*>(a.sin + o);
          if (a.mode == PE_ROPE_MIXED) {
            const f32x4 q1 = ld4(qh + i * C::LDH + 16 * nt + 4 * g);
            const f32x4 q2 = ld4(qh + i * C::LDH + 16 * (nt + C::NT / 2) + 4 * g);
            const int flat = (i - 1) * C::H + h;  // view-scramble: slot [h, i-1] holds head flat/P at pos flat%P
            const int hs = flat / P, ps = flat % P;
            const float tx = (float)(ps % a.grid), ty = (float)(ps / a.grid);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float dph = dqa[nt + C::NT / 2][r] * q1[r] - dqa[nt][r] * q2[r];
              atomicAdd(&s_dfreq[(0 * C::H + hs) * (HD / 2) + 16 * nt + 4 * g + r], tx * dph);
              atomicAdd(&s_dfreq[(1 * C::H + hs) * (HD / 2) + 16 * nt + 4 * g + r], ty * dph);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d1 = dqa[nt][r], d2 = dqa[nt + C::NT / 2][r];
            dqa[nt][r] = d1 * cs[r] + d2 * sn[r];
            dqa[nt + C::NT / 2][r] = -d1 * sn[r] + d2 * cs[r];
          }
        }
      }
      if (i < a.N) {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          st4(dq + (size_t)i * 3 * D + h * HD + 16 * dt + 4 * g, dqa[dt][0] * a.scale, dqa[dt][1] * a.scale,
              dqa[dt][2] * a.scale, dqa[dt][3] * a.scale);
      }
    }
    __syncthreads();

    // ---- step 2: (head, key tile) jobs on the plain tiles: dV, dK ---------------------------
    for (int job = wave; job < HPP * MT; job += 6) {
      const int hh = job / MT, jt = job % MT, h = h0 + hh;
      if (h >= C::H) continue;
      const T* qh = hb + (0 * HPP + hh) * C::HSZ;
      const T* kh = hb + (1 * HPP + hh) * C::HSZ;
      const T* vh = hb + (2 * HPP + hh) * C::HSZ;
      const T* doh = hb + (3 * HPP + hh) * C::HSZ;
      Frag<T> bk[C::HC], bv[C::HC];
#pragma unroll
      for (int cs = 0; cs < C::HC; ++cs) {
        bk[cs] = ld_frag(kh + (16 * jt + c) * C::LDH + 32 * cs + 8 * g);
        bv[cs] = ld_frag(vh + (16 * jt + c) * C::LDH + 32 * cs + 8 * g);
      }
      const int j = 16 * jt + c;
      f32x4 p[MT], ds[MT];
#pragma unroll
      for (int it = 0; it < MT; ++it) {
        p[it] = z4;
        ds[it] = z4;
#pragma unroll
        for (int cs = 0; cs < C::HC; ++cs) {
          mma(ld_frag(qh + (16 * it + c) * C::LDH + 32 * cs + 8 * g), bk[cs], p[it]);
          mma(ld_frag(doh + (16 * it + c) * C::LDH + 32 * cs + 8 * g), bv[cs], ds[it]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * it + 4 * g + r;
          const float lse = s_stat[(hh * 2 + 0) * C::NP + i];
          const float dl = s_stat[(hh * 2 + 1) * C::NP + i];
          const float sv = p[it][r] + pe_bias<C>(a, s_tab, s_coef, h, i, j);
          const float pv = (i < a.N && j < a.N) ? expf(sv - lse) : 0.f;
          p[it][r] = pv;
          ds[it][r] = pv * (ds[it][r] - dl);
        }
      }
      f32x4 dva[C::NT], dka[C::NT];
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt) { dva[dt] = z4; dka[dt] = z4; }
#pragma unroll
      for (int sc = 0; sc < C::SC; ++sc) {
        const bool two = (2 * sc + 1 < MT);
        const Frag<T> bp = acc_to_frag<T>(p[2 * sc], two ? p[two ? 2 * sc + 1 : 0] : z4);
        const Frag<T> bs = acc_to_frag<T>(ds[2 * sc], two ? ds[two ? 2 * sc + 1 : 0] : z4);
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt) {
          mma(ld_frag_tr(doh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bp, dva[dt]);
          mma(ld_frag_tr(qh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bs, dka[dt]);
        }
      }
      if (rope && j >= 1 && j < a.N) {
        const size_t hoff = (a.mode == PE_ROPE_MIXED) ? (size_t)h * P * (HD / 2) : 0;
#pragma unroll
        for (int nt = 0; nt < C::NT / 2; ++nt) {
          const size_t o = hoff + (size_t)(j - 1) * (HD / 2) + 16 * nt + 4 * g;
          const f32x4 cs = *reinterpret_cast<const f32x4*>(a.cos + o);
          const f32x4 sn = *reinterpret_cast<const f32x4*>(a.sin + o);
          if (a.mode == PE_ROPE_MIXED) {
            const f32x4 k1 = ld4(kh + j * C::LDH + 16 * nt + 4 * g);
            const f32x4 k2 = ld4(kh + j * C::LDH + 16 * (nt + C::NT / 2) + 4 * g);
            const int flat = (j - 1) * C::H + h;
            const int hs = flat / P, ps = flat % P;
            const float tx = (float)(ps % a.grid), ty = (float)(ps / a.grid);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float dph = dka[nt + C::NT / 2][r] * k1[r] - dka[nt][r] * k2[r];
              atomicAdd(&s_dfreq[(0 * C::H + hs) * (HD / 2) + 16 * nt + 4 * g + r], tx * dph);
              atomicAdd(&s_dfreq[(1 * C::H + hs) * (HD / 2) + 16 * nt + 4 * g + r], ty * dph);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d1 = dka[nt][r], d2 = dka[nt + C::NT / 2][r];
            dka[nt][r] = d1 * cs[r] + d2 * sn[r];
            dka[nt + C::NT / 2][r] = -d1 * sn[r] + d2 * cs[r];
          }
        }
      }
      if (j < a.N) {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt) {
          st4(dq + (size_t)j * 3 * D + D + h * HD + 16 * dt + 4 * g, dka[dt][0], dka[dt][1], dka[dt][2], dka[dt][3]);
          st4(dq + (size_t)j * 3 * D + 2 * D + h * HD + 16 * dt + 4 * g, dva[dt][0], dva[dt][1], dva[dt][2], dva[dt][3]);
        }
      }
    }
    __syncthreads();
  }

  // ---- flush this image's positional-parameter gradients ----------------------------------
  if (a.mode == PE_RELATIVE) {
    for (int q = threadIdx.x; q < C::H * (2 * a.N - 1); q += 384) {
      const int h = q / (2 * a.N - 1), i = q % (2 * a.N - 1);
      atomicAdd(a.dtable + q, s_dtab[h * C::TABLD + i]);
    }
  } else if (a.mode == PE_POLY) {
    const int nh = a.coeff_per_head ? C::H : 1;
    for (int q = threadIdx.x; q < nh * (a.degree + 1); q += 384) {
      const int h = q / (a.degree + 1), k = q % (a.degree + 1);
      atomicAdd(a.dcoeff + q, s_dcoef[h * (C::MAXDEG + 1) + k]);
    }
  } else if (a.mode == PE_ROPE_MIXED) {
    for (int q = threadIdx.x; q < 2 * C::H * (HD / 2); q += 384) atomicAdd(a.dfreqs + q, s_dfreq[q]);
  }
}

}  // namespace vitpe

using namespace vitpe;

template <typename T, int HD, int D, int MT, int HPP>
static int launch_attn(bool bwd, const AttnArgs& a, hipStream_t s) {
  if (bwd)
    hipLaunchKernelGGL((attn_bwd_kernel<T, HD, D, MT, HPP>), dim3(a.B), dim3(384), 0, s, a);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<T, HD, D, MT, HPP>), dim3(a.B), dim3(384), 0, s, a);
  VITPE_CHECK_LAUNCH();
}

static int dispatch_attn(bool bwd, int dtype, int D, int HD, const AttnArgs& a, hipStream_t s) {
  const int MT = (a.N + 15) / 16;
  if (HD != 32 || MT != 5) return (int)hipErrorNotSupported;
  if (dtype == 1) {
    if (D == 192) return launch_attn<bf16, 32, 192, 5, 2>(bwd, a, s);
    if (D == 96) return launch_attn<bf16, 32, 96, 5, 2>(bwd, a, s);
  } else if (dtype == 0) {
    if (D == 192) return launch_attn<float, 32, 192, 5, 1>(bwd, a, s);
    if (D == 96) return launch_attn<float, 32, 96, 5, 1>(bwd, a, s);
  }
  return (int)hipErrorNotSupported;
}

extern "C" int vitpe_fused_attention_supported(int dtype, int N, int D, int HD) {
  const int MT = (N + 15) / 16;
  return (dtype == 0 || dtype == 1) && HD == 32 && MT == 5 && (D == 192 || D == 96);
}

static int check_pe(int mode, const float* cos, const float* sin, const float* table, const float* coeff,
                    int N, int grid, int degree) {
  if (mode == PE_ROPE_AXIAL || mode == PE_ROPE_MIXED) {
    if (!cos || !sin || grid * grid != N - 1) return 0;
  }
  if (mode == PE_RELATIVE && !table) return 0;
  if (mode == PE_POLY && (!coeff || degree < 0 || degree > 7 || grid * grid != N - 1)) return 0;
  return mode >= PE_NONE && mode <= PE_ROPE_MIXED;
}

extern "C" int vitpe_fused_attention_fwd(int dtype, const void* xn, const void* wqkv, void* out, int B, int N,
                                         int D, int HD, int mode, const float* cos, const float* sin,
                                         const float* table, const float* coeff, int grid, int degree,
                                         int coeff_per_head, hipStream_t stream) {
  VITPE_REQUIRE(xn && wqkv && out && B >= 0 && N >= 2);
  VITPE_REQUIRE(check_pe(mode, cos, sin, table, coeff, N, grid, degree));
  if (B == 0) return 0;
  AttnArgs a{};
  a.xn = xn; a.wqkv = wqkv; a.out = out; a.cos = cos; a.sin = sin; a.table = table; a.coeff = coeff;
  a.B = B; a.N = N; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  return dispatch_attn(false, dtype, D, HD, a, stream);
}

extern "C" int vitpe_fused_attention_bwd(int dtype, const void* xn, const void* wqkv, const void* dout,
                                         void* dqkv, int B, int N, int D, int HD, int mode, const float* cos,
                                         const float* sin, const float* table, const float* coeff, int grid,
                                         int degree, int coeff_per_head, float* dtable, float* dcoeff,
                                         float* dfreqs, hipStream_t stream) {
  VITPE_REQUIRE(xn && wqkv && dout && dqkv && B >= 0 && N >= 2);
  VITPE_REQUIRE(check_pe(mode, cos, sin, table, coeff, N, grid, degree));
  if (mode == PE_RELATIVE) VITPE_REQUIRE(dtable != nullptr);
  if (mode == PE_POLY) VITPE_REQUIRE(dcoeff != nullptr);
  if (mode == PE_ROPE_MIXED) VITPE_REQUIRE(dfreqs != nullptr);
  if (B == 0) return 0;
  AttnArgs a{};
  a.xn = xn; a.wqkv = wqkv; a.out = dqkv; a.dout = dout; a.cos = cos; a.sin = sin; a.table = table;
  a.coeff = coeff; a.dtable = dtable; a.dcoeff = dcoeff; a.dfreqs = dfreqs;
  a.B = B; a.N = N; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  return dispatch_attn(true, dtype, D, HD, a, stream);
}
