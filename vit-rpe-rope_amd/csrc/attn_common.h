// Device-side pieces shared by the fused attention kernels (attn.hip: projection fused, whole
// image per workgroup) and the general attention core (attn_core.hip: one workgroup per
// (image, head), q/k/v read from a qkv buffer).
#pragma once
#include "common.h"

namespace vitpe {

struct AttnArgs {
  const void* xn;      // [B,N,D] T, layer-normed tokens
  const void* wqkv;    // attn.qkv.weight [3D,D] packed fragment-major by vitpe_pack_qkv_weights (T)
  void* out;           // fwd: [B,N,D] T merged heads ; bwd: d_qkv [B,N,3D] T
  const void* dout;    // bwd: [B,N,D] T gradient of the merged-head output
  const float* cos;    // rope: axial [P,HD/2], mixed [H,P,HD/2] (contiguous)
  const float* sin;
  const float* table;  // relative: [H,2N-1]
  const float* coeff;  // polynomial: [deg+1] or [H,deg+1]
  float* dtable;       // bwd relative: [H,2N-1] accumulated (atomics)
  float* dcoeff;       // bwd polynomial: same shape as coeff, accumulated
  float* dfreqs;       // bwd rope-mixed: [2,H,HD/2] accumulated
  const void* qkv;     // attention core (attn_core.hip): [B,N,3*H*HD] T, output of the qkv projection
  int B, N, H;
  int mode, grid, degree, coeff_per_head;
  float scale;
  // optional fused LayerNorm (reference vit.py:113,122): xn = (x - mean[row]) * rstd[row] * gamma + beta is
  // applied while staging; `xn` then points at the RAW tokens and xn_out (nullable) receives the
  // normalised tokens for the backward pass
  const float* ln_gamma;
  const float* ln_beta;
  const float* ln_mean;
  const float* ln_rstd;
  void* xn_out;
  unsigned long long* census;  // CENSUS instantiation only (vitpe_debug_attn_census): per-wave s_memtime stamps
  void* qkv_out;       // fused hd-64 forward (attn_core.hip): nullable [B,N,3*H*HD] T, the raw projection for the backward
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// kernel-level PE classes (template parameter): what the logits / gradients need
enum { KM_PLAIN = 0, KM_RELATIVE = 1, KM_POLY = 2, KM_ROPE = 3 };

template <typename T, int HD, int D, int MT, int HPP, int NTOK>
struct AttnCfg {
  static constexpr int H = D / HD;
  static constexpr int NT = HD / 16;              // 16-wide feature tiles per head
  static constexpr int KS = D / 32;               // K32 chunks of the projection
  static constexpr int HC = HD / 32;              // K32 chunks over the head dim
  static constexpr int NP = 16 * MT;              // padded tokens
  static constexpr int SC = (MT + 1) / 2;         // K32 chunks over tokens
  static constexpr int VR = 32 * SC;              // rows incl. the zero tail read by 32-deep token contractions
  // token rows: 32 B of padding -> a row stride of 26 (d=192) / 14 (d=96) 16-B slots, == 2 (mod 4): the four 16-lane
  // groups a ds_read_b128 fragment read is served in then touch 16 distinct slots each (one slot of padding, stride
  // == 1 mod 4, left them 2-way conflicting: SQ_LDS_BANK_CONFLICT was 47 % of the LDS cycles)
  static constexpr int LDX = D + (sizeof(T) == 2 ? 2 : 1) * Pad<T>::elems;   // (fp32 parity build: one slot, as validated; its LDS is full)
  static constexpr int LDH = HD + (sizeof(T) == 2 ? 2 : 1) * Pad<T>::elems;   // q/k/v/dO tiles: row stride 6 (hd 32) / 10 (hd 64) slots, == 2 (mod 4): row reads AND ds_read_b64_tr_b16 column reads conflict-free (one slot of padding: both 2-way)
  static constexpr int HSZ = VR * LDH;            // one (matrix, head) LDS tile with the zero tail
  static constexpr int QSZ = NP * LDH;            // same without the tail (row-read operands only)
  static constexpr int TABLD = 2 * NP;
  static constexpr int MAXDEG = 7;
  static constexpr int PBLD = 32;                 // polynomial bias by L1 grid distance: entries per head (grid <= 16)
  static constexpr int PESZ = H * PBLD + NP;      // s_coef: [H][PBLD] bias-by-distance (x log2 e), then NP packed (x | y << 8) token coordinates
  static constexpr int DD = D, HDD = HD, MTT = MT;
  static_assert(HD % 32 == 0 && D % HD == 0 && D % 32 == 0, "shape");
  static __device__ __forceinline__ int ntok(const AttnArgs& a) { return NTOK ? NTOK : a.N; }
};

// cross-group (lanes c, c+16, c+32, c+48) reductions on the VALU: v_permlane16_swap / 32_swap
VITPE_DEV float xg_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
VITPE_DEV float xg_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Diagonal sums of a 16x16 tile in the C layout (lane (c,g), register r <-> row 4g+r, column c): every lane of
// column c receives d0 = sum of the entries with column - row == c and d1 = sum of those with column - row == c - 16.
// Used by the relative-position table gradient, dtab[i - j + N - 1] += dS[i][j]: reducing a tile's diagonals in
// registers first (row q is rotated left by q, then the four lane groups are summed) turns 256 conflicting LDS
// atomics per tile into 31 conflict-free ones.  Uniform control flow required (cross-lane reads).
VITPE_DEV void tile_diag_sums(const f32x4& t, int lane, float& d0, float& d1) {
  const int c = lane & 15, g = lane >> 4;
  d0 = 0.f;
  d1 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rot = 4 * g + r;
    const float v = __shfl(t[r], (lane & 48) | ((c + rot) & 15), 64);   // entry (row rot, column (c + rot) & 15)
    if (c + rot >= 16) d1 += v;
    else d0 += v;
  }
  d0 = xg_sum(d0);
  d1 = xg_sum(d1);
}

// Sum over the 16 lanes of a row (lanes 16g .. 16g+15), valid in lane 15 of the row: four DPP row_shr adds on the
// VALU (no LDS traffic).  Uniform control flow required.
VITPE_DEV float row16_sum_lane15(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true));  // row_shr:8
  return v;
}

// RoPE-mixed frequency gradient of one 16-token tile (positional_encoding.py:313-351 through the view-scramble):
// lane (c, g) holds, for token tok = tok0 + c and features f0 + r (r < 4), the phase gradient dph[r]; the token's
// rotation slot [h, tok-1] carries head hs = flat / P at grid position ps = flat % P, flat = (tok-1)*H + h, so
//   dfreq[0][hs][f] += (ps % grid) * dph ,  dfreq[1][hs][f] += (ps / grid) * dph   (x scale).
// The 16 tokens of a tile map to at most (15*H)/P + 2 consecutive hs values: for each candidate the row is reduced
// with DPP adds and ONE lane per row issues the LDS atomic (the per-lane atomics were 16-way same-address conflicts).
VITPE_DEV void mixed_freq_grad_tile(float* s_dfreq, const f32x4& dph, int tok, bool tok_ok, int tok0, int h, int H, int P,
                                    int grid, int half, int f0, float scale, int lane) {
  const int c = lane & 15;
  const int flat = (max(tok, 1) - 1) * H + h;
  const int hs = flat / P, ps = flat - hs * P;
  const float tx = tok_ok ? (float)(ps % grid) * scale : 0.f, ty = tok_ok ? (float)(ps / grid) * scale : 0.f;
  const int hs_first = ((max(tok0, 1) - 1) * H + h) / P;     // wave-uniform
  const int ncand = (15 * H) / P + 2;
  for (int k = 0; k < ncand; ++k) {
    const int hsv = hs_first + k;
    if (hsv >= H) break;
    const float mx = (hs == hsv) ? tx : 0.f, my = (hs == hsv) ? ty : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sx = row16_sum_lane15(mx * dph[r]), sy = row16_sum_lane15(my * dph[r]);
      if (c == 15) {
        atomicAdd(&s_dfreq[(0 * H + hsv) * half + f0 + r], sx);
        atomicAdd(&s_dfreq[(1 * H + hsv) * half + f0 + r], sy);
      }
    }
  }
}

// L1 grid distance of patch tokens i, j >= 1 (positional_encoding.py:136-142) from the packed coordinates staged behind
// the bias table
template <typename C>
VITPE_DEV int pe_l1(const float* s_coef, int i, int j) {
  const int* xy = reinterpret_cast<const int*>(s_coef + C::H * C::PBLD);
  return (int)__builtin_amdgcn_sad_u8((unsigned)xy[i], (unsigned)xy[j], 0u);
}
// fill s_coef for KM_POLY: bias-by-distance table of heads [hbase, hbase + C::H) and the token coordinates
template <typename C>
VITPE_DEV void stage_poly(const AttnArgs& a, int hbase, float* s_coef, int N, int tid, int nthreads) {
  for (int q = tid; q < C::H * C::PBLD; q += nthreads) {
    const int h = q / C::PBLD, l = q % C::PBLD;
    const float* cf = a.coeff + (a.coeff_per_head ? (hbase + h) * (a.degree + 1) : 0);
    const float x = (float)l;
    float v = cf[a.degree];
    for (int k = a.degree - 1; k >= 0; --k) v = v * x + cf[k];
    s_coef[q] = v * LOG2E;
  }
  int* xy = reinterpret_cast<int*>(s_coef + C::H * C::PBLD);
  for (int t = tid; t < C::NP; t += nthreads) {
    const int pi = t - 1;
    xy[t] = (t >= 1 && t < N) ? ((pi % a.grid) | ((pi / a.grid) << 8)) : 0;
  }
}

// additive logit bias (already multiplied by log2 e when staged) for (query i, key j) of head h
template <typename C, int KM>
VITPE_DEV float pe_bias2(const AttnArgs& a, const float* s_tab, const float* s_coef, int h, int i, int j, int N) {
  if (KM == KM_RELATIVE) {
    int idx = i - j + N - 1;  // positional_encoding.py:67-73 (1-D index incl. class token)
    idx = max(0, min(idx, 2 * N - 2));
    return s_tab[h * C::TABLD + idx];
  }
  if (KM == KM_POLY) {
    if (i < 1 || j < 1) return 0.f;  // class row / column stay zero (positional_encoding.py:165-169)
    // polynomial of the L1 grid distance: tabulated per head and distance when the kernel starts (the distance is
    // one v_sad_u8 on packed coordinates -- the per-element integer div/mod and Horner chain were the cost of this mode)
    return s_coef[(a.coeff_per_head ? h : 0) * C::PBLD + pe_l1<C>(s_coef, i, j)];
  }
  return 0.f;
}

// ---- S^T tiles of one (head, query tile): logits in the exp2 domain, masked, + running max ----
// s[jt][r] = log2e * (scale q_i.k_j + bias(i,j)),  j = 16jt+4g+r (key), i = 16it+c (query)
template <typename T, typename C, int KM>
VITPE_DEV float logits_T(const AttnArgs& a, const T* kh, const Frag<T>* bq, const float* s_tab, const float* s_coef,
                         int h, int it, int lane, f32x4* s) {
  constexpr int MT = C::MTT;
  const int N = C::ntok(a);
  const int c = lane & 15, g = lane >> 4;
  const int i = 16 * it + c;
  const T* krow = kh + c * C::LDH + 8 * g;
  float m = -1e30f;
#pragma unroll
  for (int jt = 0; jt < MT; ++jt) {
    s[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cs = 0; cs < C::HC; ++cs) mma(ld_frag(krow + 16 * jt * C::LDH + 32 * cs), bq[cs], s[jt]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * jt + 4 * g + r;
      float v = s[jt][r];
      if (KM == KM_RELATIVE || KM == KM_POLY) v += pe_bias2<C, KM>(a, s_tab, s_coef, h, i, j, N);
      if (jt == MT - 1) v = (j < N) ? v : -1e30f;  // padding keys only exist in the last tile
      s[jt][r] = v;
      m = fmaxf(m, v);
    }
  }
  return xg_max(m);
}

}  // namespace vitpe
