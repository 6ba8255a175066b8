// General attention core for geometries whose whole-image token tile does not fit one workgroup's
// LDS (ImageNet-shaped ViT-B/16: N=197, hd=64, d=768 -- BASELINE.json config 5).
//
//   qkv = x Wqkv^T comes from vitpe_linear (panel GEMM);  this file does, per (image, head):
//   split heads -> RoPE on q,k / relative or polynomial bias -> softmax(QK^T * hd^-0.5 + bias) -> .V
//   -> merged-head output                        (reference models/vit.py:49-92, Attention.forward
//                                                 after self.qkv(x) and before self.proj)
//
// One workgroup per (image, head).  K and V of the head live in LDS (bf16 N=197: 30 + 32 KB), the
// query tile of a wave is read straight from the qkv buffer into MFMA operand registers with the
// rotation and the folded scale applied in registers -- a q row is used by exactly one wave, staging
// it through LDS would only add traffic.  The math of a (query tile) job is the one of attn.hip
// (swapped S^T = K Q^T tiles, exp2-domain softmax, accumulator-as-operand P.V).
//
// Backward keeps two LDS tiles and refills them between its two steps:
//   step 1 (query-tile jobs): K~ and V in LDS, q~/dO fragments from global: stats, dS^T, dQ
//   step 2 (key-tile jobs)  : q~ and dO in LDS, k~/v fragments from global: dV, dK
// so that fp32 (the 1e-4 parity mode) fits as well: 2 x 61 KB at N=197.
#include "attn_common.h"
#include <type_traits>

namespace vitpe {

// 16 bytes of row `rowp` (feature 0 of the head) at feature f0, rotated (rotate-half pairs
// (f, f+HD/2), rope_utils.py:85-101) with this token's cos/sin row and scaled
template <typename T, int HD, bool ROPE>
VITPE_DEV Chunk16 ld_rot_chunk(const T* rowp, int f0, const float* cs, const float* sn, bool rot, float sc) {
  constexpr int CHN = CH<T>::n;
  const Chunk16 x = *reinterpret_cast<const Chunk16*>(rowp + f0);
  if (!(ROPE && rot) && sc == 1.0f) return x;
  float f[CHN];
  chunk_to_f32<T>(x, f);
  if (ROPE && rot) {
    const bool lo = f0 < HD / 2;
    const Chunk16 y = *reinterpret_cast<const Chunk16*>(rowp + (lo ? f0 + HD / 2 : f0 - HD / 2));
    float p[CHN];
    chunk_to_f32<T>(y, p);
    const int ci = lo ? f0 : f0 - HD / 2;
    const float sg = lo ? -1.f : 1.f;
#pragma unroll
    for (int t = 0; t < CHN; ++t) f[t] = f[t] * cs[ci + t] + sg * p[t] * sn[ci + t];
  }
#pragma unroll
  for (int t = 0; t < CHN; ++t) f[t] *= sc;
  return f32_to_chunk<T>(f);
}

// K32-chunk operand fragment (8 elements at feature f0) of a global row
template <int HD, bool ROPE>
VITPE_DEV Frag<bf16> ld_rot_frag(const bf16* rowp, int f0, const float* cs, const float* sn, bool rot, float sc) {
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, ld_rot_chunk<bf16, HD, ROPE>(rowp, f0, cs, sn, rot, sc));
  return f;
}
template <int HD, bool ROPE>
VITPE_DEV Frag<float> ld_rot_frag(const float* rowp, int f0, const float* cs, const float* sn, bool rot, float sc) {
  Frag<float> f;
  const Chunk16 a = ld_rot_chunk<float, HD, ROPE>(rowp, f0, cs, sn, rot, sc);
  const Chunk16 b = ld_rot_chunk<float, HD, ROPE>(rowp, f0 + 4, cs, sn, rot, sc);
#pragma unroll
  for (int t = 0; t < 4; ++t) { f.v[t] = __uint_as_float(a[t]); f.v[4 + t] = __uint_as_float(b[t]); }
  return f;
}

// 16-B store of two adjacent [feature][token] accumulator tiles (features 16 nt0 .. 16 nt0 + 31 of the lane's token row):
// one v_permlane16_swap per dword gives every lane 8 CONTIGUOUS features (as tail2.hip's t2_store_pair)
template <bool NT = false>
VITPE_DEV void f64_store_pair(bf16* rowp, int nt0, int g, const f32x4& o0, const f32x4& o1) {
  uint32_t lo[2], hi[2];
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    bf16x2 pa, pb;
    pa[0] = (bf16)o0[2 * w2]; pa[1] = (bf16)o0[2 * w2 + 1];
    pb[0] = (bf16)o1[2 * w2]; pb[1] = (bf16)o1[2 * w2 + 1];
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, pa), __builtin_bit_cast(uint32_t, pb), false, false);
    lo[w2] = r[0]; hi[w2] = r[1];
  }
  const Chunk16 v = {lo[0], lo[1], hi[0], hi[1]};
  Chunk16* dst = reinterpret_cast<Chunk16*>(rowp + 16 * (nt0 + (g & 1)) + 8 * (g >> 1));
  if (NT) __builtin_nontemporal_store(v, dst);   // (read again only by a much later kernel)
  else *dst = v;
}

// rows of one head's matrix -> LDS tile [nrows][LDH]; rows >= N read as zero (token contractions
// run over the padded tile)
template <typename T, typename C, bool ROPE>
VITPE_DEV void stage_rows(const AttnArgs& a, const T* src, int rstride, const float* cosb, const float* sinb, float sc,
                          T* tile, int nrows, int tid, int nthreads) {
  constexpr int CHN = CH<T>::n, HD = C::HDD, CPR = HD / CHN;
  const int N = a.N;
  for (int q = tid; q < nrows * CPR; q += nthreads) {
    const int row = q / CPR, cc = q % CPR;
    Chunk16 v = {0u, 0u, 0u, 0u};
    if (row < N) {
      const int tok = max(row, 1);  // class token (row 0) is never rotated
      v = ld_rot_chunk<T, HD, ROPE>(src + (size_t)row * rstride, cc * CHN, cosb + (size_t)(tok - 1) * (HD / 2),
                                    sinb + (size_t)(tok - 1) * (HD / 2), row >= 1, sc);
    }
    *reinterpret_cast<Chunk16*>(tile + row * C::LDH + cc * CHN) = v;
  }
}

// bias table / coefficients of head hg, multiplied by log2 e (exp2-domain softmax)
template <typename C, int KM>
VITPE_DEV void stage_pe(const AttnArgs& a, int hg, float* s_tab, float* s_coef, int tid, int nthreads) {
  const int N = a.N;
  if (KM == KM_RELATIVE)
    for (int i = tid; i < C::TABLD; i += nthreads)
      s_tab[i] = (i < 2 * N - 1) ? a.table[(size_t)hg * (2 * N - 1) + i] * LOG2E : 0.f;
  if (KM == KM_POLY) stage_poly<C>(a, hg, s_coef, N, tid, nthreads);
}

// =========================================================================================
// Forward: NW = MT waves, one query tile each
// =========================================================================================
template <typename T, int HD, int MT, int KM, int NW>
__global__ __launch_bounds__(64 * NW) void attn_core_fwd_kernel(AttnArgs a) {
  using C = AttnCfg<T, HD, HD, MT, 1, 0>;
  constexpr bool ROPE = (KM == KM_ROPE);
  __shared__ __attribute__((aligned(16))) T kt[C::QSZ];  // K~ (row reads only)
  __shared__ __attribute__((aligned(16))) T vt[C::HSZ];  // V (column reads run into the zero tail)
  __shared__ __attribute__((aligned(16))) float s_tab[KM == KM_RELATIVE ? C::TABLD : 4];
  __shared__ __attribute__((aligned(16))) float s_coef[KM == KM_POLY ? C::PESZ : 4];

  const int N = a.N, H = a.H, Dr = H * HD, P = N - 1;
  const int b = blockIdx.x / H, hg = blockIdx.x % H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const T* qg = reinterpret_cast<const T*>(a.qkv) + (size_t)b * N * 3 * Dr + hg * HD;
  const size_t hoff = (ROPE && a.mode == PE_ROPE_MIXED) ? (size_t)hg * P * (HD / 2) : 0;
  const float* cosb = ROPE ? a.cos + hoff : nullptr;
  const float* sinb = ROPE ? a.sin + hoff : nullptr;

  stage_rows<T, C, ROPE>(a, qg + Dr, 3 * Dr, cosb, sinb, 1.0f, kt, C::NP, threadIdx.x, 64 * NW);
  stage_rows<T, C, false>(a, qg + 2 * Dr, 3 * Dr, nullptr, nullptr, 1.0f, vt, C::VR, threadIdx.x, 64 * NW);
  stage_pe<C, KM>(a, hg, s_tab, s_coef, threadIdx.x, 64 * NW);
  __syncthreads();

  T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * N * Dr + hg * HD;
  for (int it = wave; it < MT; it += NW) {
    const int i = 16 * it + c, il = min(i, N - 1), tok = max(il, 1);
    Frag<T> bq[C::HC];
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs)
      bq[cs] = ld_rot_frag<HD, ROPE>(qg + (size_t)il * 3 * Dr, 32 * cs + 8 * g, cosb + (size_t)(tok - 1) * (HD / 2),
                                     sinb + (size_t)(tok - 1) * (HD / 2), il >= 1, a.scale * LOG2E);
    // the rotated query fragments are FINISHED here (pinned), and the K fragment reads below stay below: unpinned, the
    // compiler hoists all 26 LDS reads above the query's global loads, carries the rotation's fp32 temporaries and spills 25
    // registers at the 128-VGPR cap of a 13-wave workgroup -- 60 MB of scratch writes per launch at the ViT-B/16 geometry
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs) pin_frag(bq[cs]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s[MT];
    const float m = logits_T<T, C, KM>(a, kt, bq, s_tab, s_coef, 0, it, lane, s);
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - m);
        s[jt][r] = p;
        l += p;
      }
    l = xg_sum(l);
    f32x4 o[C::NT];
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sc = 0; sc < C::SC; ++sc) {
      const Frag<T> bp = acc_to_frag<T>(s[2 * sc], (2 * sc + 1 < MT) ? s[(2 * sc + 1 < MT) ? 2 * sc + 1 : 0] : z4);
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt)
        mma(ld_frag_tr(vt, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bp, o[dt]);
    }
    const float inv = __builtin_amdgcn_rcpf(l);
    if (i < N) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int dt = 0; dt < C::NT; dt += 2) {
          f32x4 p0 = o[dt], p1 = o[dt + 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) { p0[r] *= inv; p1[r] *= inv; }
          f64_store_pair(reinterpret_cast<bf16*>(outp) + (size_t)i * Dr, dt, g, p0, p1);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          st4(outp + (size_t)i * Dr + 16 * dt + 4 * g, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      }
    }
  }
}

// =========================================================================================
// Fused forward at the ViT-B/16 geometry (hd = 64, N <= 208: reference vit.py:47-88 at BASELINE config 5): the head's
// slice of the qkv projection, the rotation and the core in ONE kernel -- q and k never exist outside the chip, the raw
// projection is written once (optional: the backward reads it) and never read back.
//
// One workgroup per (image, head), 9 waves.  Waves 0..6 each own 32 tokens (two 16-token tiles): their x rows come
// straight from global as B fragments (natural k order), the head's 12 weight tiles ({q, k, v} x 4 tiles of 16 output
// features) as A fragments from LDS, where waves 7 and 8 put them by LDS-DMA from the fragment-packed copy
// (vitpe_pack_weight_frags(attn.qkv.weight [3D, D], kchunk 64, phi 0): a 64-deep K chunk of one head is 24 fragments,
// 8 contiguous KB per matrix), three chunk buffers, one barrier per chunk.  A weight fragment feeds two MFMAs (the
// wave's two token tiles), so the LDS read traffic of the projection is 168 KB per chunk against 1.5 K cycles of MFMAs
// per SIMD (a wave per 16-token tile would read twice that and be LDS-bound).  The accumulators hold [feature][token]
// tiles (token on the lane): the rotate-half partner of a feature is the same register of the tile two over, so RoPE is
// in-lane; K~ goes to LDS in the k order acc_to_frag gives the q fragments (the contraction over hd does not care), V
// in natural order, q~ stays in registers as the B fragments of the wave's two query-tile jobs, which are the jobs
// of attn_core_fwd_kernel.
// =========================================================================================
constexpr int F64_CW = 7, F64_LW = 2, F64_NW = F64_CW + F64_LW;
constexpr int F64_PIECES = 24, F64_NBUF = 3;

template <int KM>
__global__ __launch_bounds__(64 * F64_NW) void attn_fused64_fwd_kernel(AttnArgs a) {
  using T = bf16;
  constexpr int HD = 64, MT = 13;
  using C = AttnCfg<T, HD, HD, MT, 1, 0>;
  constexpr bool ROPE = (KM == KM_ROPE);
  static_assert(32 * F64_CW >= C::VR, "the compute waves cover every row of the V tile");
  __shared__ __attribute__((aligned(16))) T kt[C::QSZ];
  __shared__ __attribute__((aligned(16))) T vt[C::HSZ];
  __shared__ __attribute__((aligned(16))) T wb[F64_NBUF * F64_PIECES * 512];
  __shared__ __attribute__((aligned(16))) float s_tab[KM == KM_RELATIVE ? C::TABLD : 4];
  __shared__ __attribute__((aligned(16))) float s_coef[KM == KM_POLY ? C::PESZ : 4];

  const int N = a.N, H = a.H, Dr = H * HD, P = N - 1;
  // XCD-aware order: workgroup w runs on XCD w % 8, each with its own L2.  All H heads of an image read the same x rows:
  // an XCD gets a contiguous eighth of the (image, head) list, so an image's heads share one L2 (in launch order they land
  // on eight: PMC 257 MB fetched per launch at B = 64 for 97 MB of operands).
  int wg = (int)blockIdx.x;
  if ((gridDim.x & 7) == 0) wg = (wg & 7) * (int)(gridDim.x >> 3) + (wg >> 3);
  const int b = wg / H, hg = wg % H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nchunk = Dr / 64;
  const size_t hoff = (ROPE && a.mode == PE_ROPE_MIXED) ? (size_t)hg * P * (HD / 2) : 0;
  const float* cosb = ROPE ? a.cos + hoff : nullptr;
  const float* sinb = ROPE ? a.sin + hoff : nullptr;

  stage_pe<C, KM>(a, hg, s_tab, s_coef, threadIdx.x, 64 * F64_NW);

  if (wave >= F64_CW) {
    // ---- loader waves: 12 fragments each per chunk (wave 7: q and half of k; wave 8: the rest) ---------------------------
    const int lw = wave - F64_CW;
    const T* const wsrc = reinterpret_cast<const T*>(a.wqkv) + lane * 8;
    const int TD = Dr / 16;                      // 16-row tiles per matrix; 3 TD per K chunk in the packed copy
    auto dma = [&](int kc, int buf) {
#pragma unroll
      for (int i = 0; i < F64_PIECES / F64_LW; ++i) {
        const int p = (F64_PIECES / F64_LW) * lw + i, m = p >> 3, q8 = p & 7;
        const T* src = wsrc + ((size_t)(kc * 3 * TD + m * TD + 4 * hg) * 2 + q8) * 512;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(wb + (buf * F64_PIECES + p) * 512), 16, 0, 0);
      }
    };
    dma(0, 0);
    if (nchunk > 1) dma(1, 1);
    for (int kc = 0; kc < nchunk; ++kc) {
      if (kc + 1 < nchunk) __builtin_amdgcn_s_waitcnt(0x0F7C);   // vmcnt(12): chunk kc has landed, chunk kc + 1 may be in flight
      else __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
      // (a bare s_barrier: __syncthreads() would drain vmcnt and with it the chunk in flight)
      asm volatile("s_barrier" ::: "memory");                    // chunk kc visible; everybody has left chunk kc - 1's buffer
      if (kc + 2 < nchunk) dma(kc + 2, (kc + 2) % F64_NBUF);
    }
    __syncthreads();                                             // (the barrier behind the K~ / V tiles)
    return;
  }

  // ---- compute waves: projection of this wave's 32 tokens -----------------------------------------------------------------
  int tok[2];
  const T* xrow[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    tok[tt] = 32 * wave + 16 * tt + c;
    xrow[tt] = reinterpret_cast<const T*>(a.xn) + ((size_t)b * N + min(tok[tt], N - 1)) * Dr + 8 * g;
  }
  f32x4 acc[3][4][2];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[m][j][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag<T> cur[2][2], nxt[2][2];                  // [token tile][k step] of the chunk in work / the next one
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) nxt[tt][ks] = ld_frag(xrow[tt] + 32 * ks);
  for (int kc = 0; kc < nchunk; ++kc) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) cur[tt][ks] = nxt[tt][ks];
    asm volatile("s_barrier" ::: "memory");      // (this wave's reads of chunk kc - 1 fed MFMAs already issued: nothing to drain)
    if (kc + 1 < nchunk) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) nxt[tt][ks] = ld_frag(xrow[tt] + 64 * (kc + 1) + 32 * ks);
    }
    const T* wf = wb + (kc % F64_NBUF) * F64_PIECES * 512 + lane * 8;
    // 24 weight fragments per chunk in groups of three through two register sets: group n + 1 is read while group n feeds
    // its six MFMAs (left alone the scheduler hoists all 24 reads to the top: 96 registers, 160 spilled)
    {
      constexpr int GS = 3, NG = 24 / GS;
      Frag<T> wr[2][GS];
      auto rd = [&](int gi, Frag<T> (&dst)[GS]) {
#pragma unroll
        for (int t = 0; t < GS; ++t) {
          const int f = gi * GS + t, ks = f / 12, mj = f % 12;
          dst[t] = ld_frag(wf + (mj * 2 + ks) * 512);
        }
      };
      rd(0, wr[0]);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) rd(gi + 1, wr[(gi + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < GS; ++t) {
          const int f = gi * GS + t, ks = f / 12, mj = f % 12;
          mma(wr[gi & 1][t], cur[0][ks], acc[mj >> 2][mj & 3][0]);
          mma(wr[gi & 1][t], cur[1][ks], acc[mj >> 2][mj & 3][1]);   // (the last wave's second tile is padding: computed anyway --
                                                                     //  a second copy of this loop behind a branch cost 230 spills)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- raw projection out (the values every later step sees are the bf16-rounded ones, as on the unfused path) ----------
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][j][tt][r] = to_f32(from_f32<T>(acc[m][j][tt][r]));
  if (a.qkv_out != nullptr) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      // (the swaps pair lane groups: every lane takes part, rows past N write their clamped row's address -- row N - 1,
      //  same values as that row's own lane: x was read from the clamped row)
      T* qo = reinterpret_cast<T*>(a.qkv_out) + ((size_t)b * N + min(tok[tt], N - 1)) * 3 * Dr + hg * HD;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int j = 0; j < 4; j += 2) f64_store_pair<true>(qo + (size_t)m * Dr, j, g, acc[m][j][tt], acc[m][j + 1][tt]);
    }
  }
  // ---- rotation of q and k (rotate-half pairs (f, f + 32): tile j and tile j + 2, same register), class token excluded -----
  if (ROPE) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tok[tt] >= 1 && tok[tt] < N) {
        const float* csr = cosb + (size_t)(tok[tt] - 1) * (HD / 2) + 4 * g;
        const float* snr = sinb + (size_t)(tok[tt] - 1) * (HD / 2) + 4 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(csr + 16 * j), sv = *reinterpret_cast<const f32x4*>(snr + 16 * j);
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float lo = acc[m][j][tt][r], hi = acc[m][j + 2][tt][r];
              acc[m][j][tt][r] = lo * cv[r] - hi * sv[r];
              acc[m][j + 2][tt][r] = hi * cv[r] + lo * sv[r];
            }
        }
      }
    }
  }
  // ---- K~ (k order of the q fragments) and V (natural) -> LDS, rows past N zero; q~ -> B fragments ---------------------------
  Frag<T> bq[2][C::HC];
  const float qsc = a.scale * LOG2E;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int row = 32 * wave + 16 * tt + c;
    const float live = tok[tt] < N ? 1.f : 0.f;
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs) {
      f32x4 klo = acc[1][2 * cs][tt], khi = acc[1][2 * cs + 1][tt], qlo = acc[0][2 * cs][tt], qhi = acc[0][2 * cs + 1][tt];
#pragma unroll
      for (int r = 0; r < 4; ++r) { klo[r] *= live; khi[r] *= live; qlo[r] *= qsc; qhi[r] *= qsc; }
      if (row < C::NP) *reinterpret_cast<bf16x8*>(kt + row * C::LDH + 32 * cs + 8 * g) = acc_to_frag<T>(klo, khi).v;
      bq[tt][cs] = acc_to_frag<T>(qlo, qhi);
      pin_frag(bq[tt][cs]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      st4(vt + row * C::LDH + 16 * j + 4 * g, acc[2][j][tt][0] * live, acc[2][j][tt][1] * live, acc[2][j][tt][2] * live,
          acc[2][j][tt][3] * live);
  }
  __syncthreads();

  // ---- the core: this wave's two query tiles (attn_core_fwd_kernel's job) -------------------------------------------------
  T* outp = reinterpret_cast<T*>(a.out) + (size_t)b * N * Dr + hg * HD;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int it = 2 * wave + tt;
    if (it >= MT) break;
    const int i = 16 * it + c;
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s[MT];
    const float mx = logits_T<T, C, KM>(a, kt, bq[tt], s_tab, s_coef, 0, it, lane, s);
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - mx);
        s[jt][r] = p;
        l += p;
      }
    l = xg_sum(l);
    f32x4 o[C::NT];
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sc = 0; sc < C::SC; ++sc) {
      const Frag<T> bp = acc_to_frag<T>(s[2 * sc], (2 * sc + 1 < MT) ? s[(2 * sc + 1 < MT) ? 2 * sc + 1 : 0] : z4);
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt)
        mma(ld_frag_tr(vt, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bp, o[dt]);
    }
    const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= inv;
    if (i < N) {   // (the swap pairs lanes of the same token: both sides of a pair are in or out together)
#pragma unroll
      for (int dt = 0; dt < C::NT; dt += 2) f64_store_pair(outp + (size_t)i * Dr, dt, g, o[dt], o[dt + 1]);
    }
  }
}

// =========================================================================================
// Backward
// =========================================================================================
constexpr int CORE_HMAX = 16;  // heads, for the RoPE-mixed frequency-gradient scratch

template <typename T, int HD, int MT, int KM, int NW>
__global__ __launch_bounds__(64 * NW) void attn_core_bwd_kernel(AttnArgs a) {
  using C = AttnCfg<T, HD, HD, MT, 1, 0>;
  constexpr bool ROPE = (KM == KM_ROPE);
  constexpr int NTH = 64 * NW;
  __shared__ __attribute__((aligned(16))) T t0[C::HSZ];  // step 1: K~ ; step 2: q~
  __shared__ __attribute__((aligned(16))) T t1[C::HSZ];  // step 1: V  ; step 2: dO
  __shared__ __attribute__((aligned(16))) float s_tab[KM == KM_RELATIVE ? C::TABLD : 4];
  __shared__ __attribute__((aligned(16))) float s_coef[KM == KM_POLY ? C::PESZ : 4];
  __shared__ __attribute__((aligned(16))) float s_stat[2 * C::NP];  // [lse2 | delta][token]
  __shared__ float s_dtab[KM == KM_RELATIVE ? C::TABLD : 4];
  __shared__ float s_dcoef[C::MAXDEG + 1];
  __shared__ float s_dfreq[ROPE ? 2 * CORE_HMAX * (HD / 2) : 4];

  const int N = a.N, H = a.H, Dr = H * HD, P = N - 1;
  const int b = blockIdx.x / H, hg = blockIdx.x % H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const T* qg = reinterpret_cast<const T*>(a.qkv) + (size_t)b * N * 3 * Dr + hg * HD;
  const T* dog = reinterpret_cast<const T*>(a.dout) + (size_t)b * N * Dr + hg * HD;
  T* dq = reinterpret_cast<T*>(a.out) + (size_t)b * N * 3 * Dr + hg * HD;
  const bool mixed = ROPE && a.mode == PE_ROPE_MIXED;
  const size_t hoff = mixed ? (size_t)hg * P * (HD / 2) : 0;
  const float* cosb = ROPE ? a.cos + hoff : nullptr;
  const float* sinb = ROPE ? a.sin + hoff : nullptr;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const float qsc = a.scale * LOG2E;

  stage_rows<T, C, ROPE>(a, qg + Dr, 3 * Dr, cosb, sinb, 1.0f, t0, C::VR, threadIdx.x, NTH);
  stage_rows<T, C, false>(a, qg + 2 * Dr, 3 * Dr, nullptr, nullptr, 1.0f, t1, C::NP, threadIdx.x, NTH);
  stage_pe<C, KM>(a, hg, s_tab, s_coef, threadIdx.x, NTH);
  if (KM == KM_RELATIVE)
    for (int q = threadIdx.x; q < C::TABLD; q += NTH) s_dtab[q] = 0.f;
  for (int q = threadIdx.x; q <= C::MAXDEG; q += NTH) s_dcoef[q] = 0.f;
  if (ROPE)
    for (int q = threadIdx.x; q < 2 * CORE_HMAX * (HD / 2); q += NTH) s_dfreq[q] = 0.f;
  __syncthreads();

  // ---- step 1: query-tile jobs on the swapped tiles: stats, dS^T, dQ -----------------------
  for (int it = wave; it < MT; it += NW) {
    const T* kh = t0;
    const T* vh = t1;
    const int i = 16 * it + c, il = min(i, N - 1), tok = max(il, 1);
    const float* csr = cosb + (size_t)(tok - 1) * (HD / 2);
    const float* snr = sinb + (size_t)(tok - 1) * (HD / 2);
    Frag<T> bq[C::HC], bdo[C::HC];
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs) {
      bq[cs] = ld_rot_frag<HD, ROPE>(qg + (size_t)il * 3 * Dr, 32 * cs + 8 * g, csr, snr, il >= 1, qsc);
      bdo[cs] = ld_rot_frag<HD, false>(dog + (size_t)il * Dr, 32 * cs + 8 * g, nullptr, nullptr, false, 1.0f);
    }
    f32x4 s[MT], dp[MT];
    const float m = logits_T<T, C, KM>(a, kh, bq, s_tab, s_coef, 0, it, lane, s);
    if (MT > 8) __builtin_amdgcn_sched_barrier(0);
    const T* vrow = vh + c * C::LDH + 8 * g;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt) {
      dp[jt] = z4;
#pragma unroll
      for (int cs = 0; cs < C::HC; ++cs) mma(ld_frag(vrow + 16 * jt * C::LDH + 32 * cs), bdo[cs], dp[jt]);
      if (MT > 8 && (jt & 1)) __builtin_amdgcn_sched_barrier(0);  // bound the load hoisting (register pressure)
    }
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[jt][r] - m);
        s[jt][r] = p;
        l += p;
      }
    l = xg_sum(l);
    const float inv = __builtin_amdgcn_rcpf(l);
    float dl = 0.f;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[jt][r] *= inv;
        dl += s[jt][r] * dp[jt][r];
      }
    dl = xg_sum(dl);
    if (g == 0) {
      s_stat[0 * C::NP + i] = m + __builtin_amdgcn_logf(l);  // v_log_f32 = log2
      s_stat[1 * C::NP + i] = dl;
    }
    float cacc[C::MAXDEG + 1];
#pragma unroll
    for (int k = 0; k <= C::MAXDEG; ++k) cacc[k] = 0.f;
    const bool qvalid = i < N;
#pragma unroll
    for (int jt = 0; jt < MT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * g + r;
        const bool valid = qvalid && ((jt < MT - 1) || (j < N));
        const float ds = valid ? s[jt][r] * (dp[jt][r] - dl) : 0.f;
        dp[jt][r] = ds;
        if (KM == KM_POLY) {
          if (valid && i >= 1 && j >= 1) {
            const float x = (float)pe_l1<C>(s_coef, i, j);
            float pw = 1.f;
#pragma unroll
            for (int k = 0; k <= C::MAXDEG; ++k) {
              if (k <= a.degree) cacc[k] += ds * pw;
              pw *= x;
            }
          }
        }
      }
    if (KM == KM_RELATIVE) {   // diagonals of every 16x16 tile reduced in registers, then conflict-free LDS atomics
#pragma unroll
      for (int jt = 0; jt < MT; ++jt) {
        float d0, d1;
        tile_diag_sums(dp[jt], lane, d0, d1);
        const int idx0 = 16 * (it - jt) + c + N - 1;
        if (g == 0) {
          if (idx0 >= 0 && idx0 <= 2 * N - 2) atomicAdd(&s_dtab[idx0], d0);
          if (c >= 1 && idx0 - 16 >= 0 && idx0 - 16 <= 2 * N - 2) atomicAdd(&s_dtab[idx0 - 16], d1);
        }
      }
    }
    if (KM == KM_POLY) {
#pragma unroll
      for (int k = 0; k <= C::MAXDEG; ++k) {
        if (k <= a.degree) {  // wave-uniform
          const float t = wave_sum(cacc[k]);
          if (lane == 0) atomicAdd(&s_dcoef[k], t);
        }
      }
    }
    // dQrot^T[d][i] / scale = sum_j K~^T[d][j] dS^T[j][i]
    f32x4 dqa[C::NT];
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt) dqa[dt] = z4;
#pragma unroll
    for (int sc = 0; sc < C::SC; ++sc) {
      const Frag<T> bs = acc_to_frag<T>(dp[2 * sc], (2 * sc + 1 < MT) ? dp[(2 * sc + 1 < MT) ? 2 * sc + 1 : 0] : z4);
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt)
        mma(ld_frag_tr(kh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bs, dqa[dt]);
      if (MT > 8) __builtin_amdgcn_sched_barrier(0);
    }
    if (mixed) {   // uniform branch: row reductions inside (mixed_freq_grad_tile, attn_common.h)
      // dL/dphase = (dq~2 q~1 - dq~1 q~2), q~ = scale*log2e*rot(q) (see attn.hip): ln2 undoes the log2e
      const bool tok_ok = i >= 1 && i < N;
#pragma unroll
      for (int nt = 0; nt < C::NT / 2; ++nt) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(csr + 16 * nt + 4 * g);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(snr + 16 * nt + 4 * g);
        const f32x4 x1 = ld4(qg + (size_t)il * 3 * Dr + 16 * nt + 4 * g);
        const f32x4 x2 = ld4(qg + (size_t)il * 3 * Dr + 16 * (nt + C::NT / 2) + 4 * g);
        f32x4 dph;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float q1 = (x1[r] * cs[r] - x2[r] * sn[r]) * qsc, q2 = (x1[r] * sn[r] + x2[r] * cs[r]) * qsc;
          dph[r] = dqa[nt + C::NT / 2][r] * q1 - dqa[nt][r] * q2;
        }
        mixed_freq_grad_tile(s_dfreq, dph, i, tok_ok, 16 * it, hg, H, P, a.grid, HD / 2, 16 * nt + 4 * g, LN2, lane);
      }
    }
    if (ROPE && i >= 1 && i < N) {
#pragma unroll
      for (int nt = 0; nt < C::NT / 2; ++nt) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(csr + 16 * nt + 4 * g);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(snr + 16 * nt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d1 = dqa[nt][r], d2 = dqa[nt + C::NT / 2][r];
          dqa[nt][r] = d1 * cs[r] + d2 * sn[r];
          dqa[nt + C::NT / 2][r] = -d1 * sn[r] + d2 * cs[r];
        }
      }
    }
    if (i < N) {
      if constexpr (sizeof(T) == 2) {   // 16-B pieces (lanes of one token pair up: both sides of a swap pass the guard together)
#pragma unroll
        for (int dt = 0; dt < C::NT; dt += 2) {
          f32x4 p0 = dqa[dt], p1 = dqa[dt + 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) { p0[r] *= a.scale; p1[r] *= a.scale; }
          f64_store_pair(reinterpret_cast<bf16*>(dq) + (size_t)i * 3 * Dr, dt, g, p0, p1);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt)
          st4(dq + (size_t)i * 3 * Dr + 16 * dt + 4 * g, dqa[dt][0] * a.scale, dqa[dt][1] * a.scale, dqa[dt][2] * a.scale,
              dqa[dt][3] * a.scale);
      }
    }
  }
  __syncthreads();
  // refill: q~ and dO with zero tails (token contractions of step 2)
  stage_rows<T, C, ROPE>(a, qg, 3 * Dr, cosb, sinb, qsc, t0, C::VR, threadIdx.x, NTH);
  stage_rows<T, C, false>(a, dog, Dr, nullptr, nullptr, 1.0f, t1, C::VR, threadIdx.x, NTH);
  __syncthreads();

  // ---- step 2: key-tile jobs on the plain tiles: dV, dK -------------------------------------
  for (int jt = wave; jt < MT; jt += NW) {
    const T* qh = t0;
    const T* doh = t1;
    const int j = 16 * jt + c, jl = min(j, N - 1), tok = max(jl, 1);
    const float* csr = cosb + (size_t)(tok - 1) * (HD / 2);
    const float* snr = sinb + (size_t)(tok - 1) * (HD / 2);
    Frag<T> bk[C::HC], bv[C::HC];
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs) {
      bk[cs] = ld_rot_frag<HD, ROPE>(qg + Dr + (size_t)jl * 3 * Dr, 32 * cs + 8 * g, csr, snr, jl >= 1, 1.0f);
      bv[cs] = ld_rot_frag<HD, false>(qg + 2 * Dr + (size_t)jl * 3 * Dr, 32 * cs + 8 * g, nullptr, nullptr, false, 1.0f);
    }
    const bool kvalid = j < N;
    const T* qrow = qh + c * C::LDH + 8 * g;
    const T* dorow = doh + c * C::LDH + 8 * g;
    f32x4 dva[C::NT], dka[C::NT];
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt) { dva[dt] = z4; dka[dt] = z4; }
    // the row statistics are known here, so the query tiles stream through in pairs (one K32 chunk
    // of the token contraction): only two P / dS tiles are live at a time
#pragma unroll
    for (int sc = 0; sc < C::SC; ++sc) {
      f32x4 p[2], ds[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int it = 2 * sc + hf;
        p[hf] = z4;
        ds[hf] = z4;
        if (it < MT) {
#pragma unroll
          for (int cs = 0; cs < C::HC; ++cs) {
            mma(ld_frag(qrow + 16 * it * C::LDH + 32 * cs), bk[cs], p[hf]);
            mma(ld_frag(dorow + 16 * it * C::LDH + 32 * cs), bv[cs], ds[hf]);
          }
          const f32x4 lse = *reinterpret_cast<const f32x4*>(&s_stat[0 * C::NP + 16 * it + 4 * g]);
          const f32x4 dl = *reinterpret_cast<const f32x4*>(&s_stat[1 * C::NP + 16 * it + 4 * g]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * g + r;
            float sv = p[hf][r];
            if (KM == KM_RELATIVE || KM == KM_POLY) sv += pe_bias2<C, KM>(a, s_tab, s_coef, 0, i, j, N);
            const bool valid = kvalid && ((it < MT - 1) || (i < N));
            const float pv = valid ? __builtin_amdgcn_exp2f(sv - lse[r]) : 0.f;
            p[hf][r] = pv;
            ds[hf][r] = pv * (ds[hf][r] - dl[r]);
          }
        }
      }
      const Frag<T> bp = acc_to_frag<T>(p[0], p[1]);
      const Frag<T> bs = acc_to_frag<T>(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < C::NT; ++dt) {
        mma(ld_frag_tr(doh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bp, dva[dt]);
        mma(ld_frag_tr(qh, C::LDH, 32 * sc + 4 * g, 32 * sc + 16 + 4 * g, 16 * dt), bs, dka[dt]);
      }
      if (MT > 8) __builtin_amdgcn_sched_barrier(0);
    }
    // dK_rot = dS^T q~ / log2e  (q~ carries the folded scale and log2e)
#pragma unroll
    for (int dt = 0; dt < C::NT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dka[dt][r] *= LN2;
    if (mixed) {
      const bool tok_ok = j >= 1 && j < N;
#pragma unroll
      for (int nt = 0; nt < C::NT / 2; ++nt) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(csr + 16 * nt + 4 * g);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(snr + 16 * nt + 4 * g);
        const f32x4 x1 = ld4(qg + Dr + (size_t)jl * 3 * Dr + 16 * nt + 4 * g);
        const f32x4 x2 = ld4(qg + Dr + (size_t)jl * 3 * Dr + 16 * (nt + C::NT / 2) + 4 * g);
        f32x4 dph;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float k1 = x1[r] * cs[r] - x2[r] * sn[r], k2 = x1[r] * sn[r] + x2[r] * cs[r];
          dph[r] = dka[nt + C::NT / 2][r] * k1 - dka[nt][r] * k2;
        }
        mixed_freq_grad_tile(s_dfreq, dph, j, tok_ok, 16 * jt, hg, H, P, a.grid, HD / 2, 16 * nt + 4 * g, 1.0f, lane);
      }
    }
    if (ROPE && j >= 1 && j < N) {
#pragma unroll
      for (int nt = 0; nt < C::NT / 2; ++nt) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(csr + 16 * nt + 4 * g);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(snr + 16 * nt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d1 = dka[nt][r], d2 = dka[nt + C::NT / 2][r];
          dka[nt][r] = d1 * cs[r] + d2 * sn[r];
          dka[nt + C::NT / 2][r] = -d1 * sn[r] + d2 * cs[r];
        }
      }
    }
    if (j < N) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int dt = 0; dt < C::NT; dt += 2) {
          f64_store_pair(reinterpret_cast<bf16*>(dq) + Dr + (size_t)j * 3 * Dr, dt, g, dka[dt], dka[dt + 1]);
          f64_store_pair(reinterpret_cast<bf16*>(dq) + 2 * Dr + (size_t)j * 3 * Dr, dt, g, dva[dt], dva[dt + 1]);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < C::NT; ++dt) {
          st4(dq + Dr + (size_t)j * 3 * Dr + 16 * dt + 4 * g, dka[dt][0], dka[dt][1], dka[dt][2], dka[dt][3]);
          st4(dq + 2 * Dr + (size_t)j * 3 * Dr + 16 * dt + 4 * g, dva[dt][0], dva[dt][1], dva[dt][2], dva[dt][3]);
        }
      }
    }
  }
  __syncthreads();

  // ---- flush this (image, head)'s positional-parameter gradients ----------------------------
  if (KM == KM_RELATIVE) {
    for (int q = threadIdx.x; q < 2 * N - 1; q += NTH) atomicAdd(a.dtable + (size_t)hg * (2 * N - 1) + q, s_dtab[q]);
  } else if (KM == KM_POLY) {
    for (int k = threadIdx.x; k <= a.degree; k += NTH)
      atomicAdd(a.dcoeff + (a.coeff_per_head ? hg * (a.degree + 1) : 0) + k, s_dcoef[k]);
  } else if (mixed) {
    for (int q = threadIdx.x; q < 2 * H * (HD / 2); q += NTH)
      if (s_dfreq[q] != 0.f) atomicAdd(a.dfreqs + q, s_dfreq[q]);
  }
}

}  // namespace vitpe

using namespace vitpe;

template <typename T, int HD, int MT>
static int launch_core(bool bwd, const AttnArgs& a, hipStream_t s) {
  // forward: one wave per query tile (two rounds above 13 tiles); backward: half as many waves (two rounds) so that
  // the p/dS accumulators of a key-tile job (2*MT f32x4) stay in registers
  constexpr int NWF = (MT > 13) ? (MT + 1) / 2 : MT, NWB = (MT > 8) ? (MT + 1) / 2 : MT;
  const dim3 grid((unsigned)(a.B * a.H));
#define VITPE_CORE_LAUNCH(KM)                                                                                      \
  do {                                                                                                               \
    if (bwd) hipLaunchKernelGGL((attn_core_bwd_kernel<T, HD, MT, KM, NWB>), grid, dim3(64 * NWB), 0, s, a);          \
    else hipLaunchKernelGGL((attn_core_fwd_kernel<T, HD, MT, KM, NWF>), grid, dim3(64 * NWF), 0, s, a);              \
  } while (0)
  switch (a.mode) {
    case PE_RELATIVE: VITPE_CORE_LAUNCH(KM_RELATIVE); break;
    case PE_POLY: VITPE_CORE_LAUNCH(KM_POLY); break;
    case PE_ROPE_AXIAL:
    case PE_ROPE_MIXED: VITPE_CORE_LAUNCH(KM_ROPE); break;
    default: VITPE_CORE_LAUNCH(KM_PLAIN); break;
  }
#undef VITPE_CORE_LAUNCH
  VITPE_CHECK_LAUNCH();
}

// Instantiated geometries: head dimension 32 / 64 and these token-tile counts MT = ceil(N / 16) -- the square grids the
// reference CLI can produce from its --img_size / --patch_size flags ((img/patch)^2 + 1 tokens): N = 17 (32/8),
// 50 (28/4, 224/32), 65 (32/4, 64/8: also the fused path), 145..160 (48/4), 197 (224/16: config 5), 257 (64/4, 32/2).
// Both LDS tiles of the backward must fit 160 KB: fp32 at hd = 64 stops at 13 tiles.
template <typename T, int HD, int MT>
constexpr bool core_fits() {
  using C = AttnCfg<T, HD, HD, MT, 1, 0>;
  return 2 * (size_t)C::HSZ * sizeof(T) + 6 * 1024 <= 160 * 1024 && MT <= 17;
}
#define VITPE_CORE_MTS(X, T, HD) X(T, HD, 2) X(T, HD, 4) X(T, HD, 5) X(T, HD, 10) X(T, HD, 13) X(T, HD, 17)

template <typename T, int HD>
static int dispatch_core_t(bool bwd, int MT, const AttnArgs& a, hipStream_t s) {
#define VITPE_CORE_CASE(T_, HD_, MT_)                                                    \
  if (MT == MT_) {                                                                       \
    if constexpr (core_fits<T_, HD_, MT_>()) return launch_core<T_, HD_, MT_>(bwd, a, s); \
    else return (int)hipErrorNotSupported;                                               \
  }
  VITPE_CORE_MTS(VITPE_CORE_CASE, T, HD)
#undef VITPE_CORE_CASE
  return (int)hipErrorNotSupported;
}

template <typename T, int HD>
static bool core_supported_t(int MT) {
#define VITPE_CORE_CASE(T_, HD_, MT_) if (MT == MT_) return core_fits<T_, HD_, MT_>();
  VITPE_CORE_MTS(VITPE_CORE_CASE, T, HD)
#undef VITPE_CORE_CASE
  return false;
}

static int dispatch_core(bool bwd, int dtype, int HD, const AttnArgs& a, hipStream_t s) {
  const int MT = (a.N + 15) / 16;
  if (dtype == 1 && HD == 64) return dispatch_core_t<bf16, 64>(bwd, MT, a, s);
  if (dtype == 1 && HD == 32) return dispatch_core_t<bf16, 32>(bwd, MT, a, s);
  if (dtype == 0 && HD == 64) return dispatch_core_t<float, 64>(bwd, MT, a, s);
  if (dtype == 0 && HD == 32) return dispatch_core_t<float, 32>(bwd, MT, a, s);
  return (int)hipErrorNotSupported;
}

extern "C" int vitpe_attention_core_supported(int dtype, int N, int HD) {
  const int MT = (N + 15) / 16;
  if (N < 2) return 0;
  if (dtype == 1 && HD == 64) return core_supported_t<bf16, 64>(MT);
  if (dtype == 1 && HD == 32) return core_supported_t<bf16, 32>(MT);
  if (dtype == 0 && HD == 64) return core_supported_t<float, 64>(MT);
  if (dtype == 0 && HD == 32) return core_supported_t<float, 32>(MT);
  return 0;
}

static int core_check_pe(int mode, const float* cos, const float* sin, const float* table, const float* coeff, int N,
                         int H, int grid, int degree) {
  if (mode == PE_ROPE_AXIAL || mode == PE_ROPE_MIXED) {
    if (!cos || !sin || grid * grid != N - 1) return 0;
    if (mode == PE_ROPE_MIXED && H > CORE_HMAX) return 0;
  }
  if (mode == PE_RELATIVE && !table) return 0;
  if (mode == PE_POLY && (!coeff || degree < 0 || degree > 7 || grid * grid != N - 1)) return 0;
  return mode >= PE_NONE && mode <= PE_ROPE_MIXED;
}

extern "C" int vitpe_attention_core_fwd(int dtype, const void* qkv, void* out, int B, int N, int H, int HD, int mode,
                                        const float* cos, const float* sin, const float* table, const float* coeff,
                                        int grid, int degree, int coeff_per_head, hipStream_t stream) {
  VITPE_REQUIRE(qkv && out && B >= 0 && N >= 2 && H >= 1);
  VITPE_REQUIRE(core_check_pe(mode, cos, sin, table, coeff, N, H, grid, degree));
  if (B == 0) return 0;
  AttnArgs a{};
  a.qkv = qkv; a.out = out; a.cos = cos; a.sin = sin; a.table = table; a.coeff = coeff;
  a.B = B; a.N = N; a.H = H; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  return dispatch_core(false, dtype, HD, a, stream);
}

// The fused forward at hd = 64 (attn_fused64_fwd_kernel): bf16, 193 <= N <= 208 (13 token tiles), H <= 16 heads of 64.
extern "C" int vitpe_attention_fused64_supported(int dtype, int N, int H, int HD) {
  return dtype == 1 && HD == 64 && (N + 15) / 16 == 13 && H >= 1 && H <= CORE_HMAX;
}

extern "C" int vitpe_attention_fused64_fwd(int dtype, const void* xn, const void* wqkv_packed, void* qkv_out, void* out, int B,
                                           int N, int H, int HD, int mode, const float* cos, const float* sin,
                                           const float* table, const float* coeff, int grid, int degree, int coeff_per_head,
                                           hipStream_t stream) {
  VITPE_REQUIRE(xn && wqkv_packed && out && B >= 0);
  if (!vitpe_attention_fused64_supported(dtype, N, H, HD)) return (int)hipErrorNotSupported;
  VITPE_REQUIRE(core_check_pe(mode, cos, sin, table, coeff, N, H, grid, degree));
  if (B == 0) return 0;
  AttnArgs a{};
  a.xn = xn; a.wqkv = wqkv_packed; a.qkv_out = qkv_out; a.out = out; a.cos = cos; a.sin = sin; a.table = table; a.coeff = coeff;
  a.B = B; a.N = N; a.H = H; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  const dim3 grid_((unsigned)(B * H)), block(64 * F64_NW);
  switch (mode) {
    case PE_RELATIVE: hipLaunchKernelGGL((attn_fused64_fwd_kernel<KM_RELATIVE>), grid_, block, 0, stream, a); break;
    case PE_POLY: hipLaunchKernelGGL((attn_fused64_fwd_kernel<KM_POLY>), grid_, block, 0, stream, a); break;
    case PE_ROPE_AXIAL:
    case PE_ROPE_MIXED: hipLaunchKernelGGL((attn_fused64_fwd_kernel<KM_ROPE>), grid_, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((attn_fused64_fwd_kernel<KM_PLAIN>), grid_, block, 0, stream, a); break;
  }
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_attention_core_bwd(int dtype, const void* qkv, const void* dout, void* dqkv, int B, int N, int H,
                                        int HD, int mode, const float* cos, const float* sin, const float* table,
                                        const float* coeff, int grid, int degree, int coeff_per_head, float* dtable,
                                        float* dcoeff, float* dfreqs, hipStream_t stream) {
  VITPE_REQUIRE(qkv && dout && dqkv && B >= 0 && N >= 2 && H >= 1);
  VITPE_REQUIRE(core_check_pe(mode, cos, sin, table, coeff, N, H, grid, degree));
  if (mode == PE_RELATIVE) VITPE_REQUIRE(dtable);
  if (mode == PE_POLY) VITPE_REQUIRE(dcoeff);
  if (mode == PE_ROPE_MIXED) VITPE_REQUIRE(dfreqs);
  if (B == 0) return 0;
  AttnArgs a{};
  a.qkv = qkv; a.dout = dout; a.out = dqkv; a.cos = cos; a.sin = sin; a.table = table; a.coeff = coeff;
  a.dtable = dtable; a.dcoeff = dcoeff; a.dfreqs = dfreqs;
  a.B = B; a.N = N; a.H = H; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  return dispatch_core(true, dtype, HD, a, stream);
}
