// Block tail ("wave per token tile"): the attention projection + residual + LayerNorm2 + the whole
// MLP branch + residual of a transformer block (reference models/vit.py:91,116-118,122-124; timm Mlp = fc1 -> GELU ->
// fc2) in one kernel in which the hidden activation NEVER leaves the registers between fc1 and fc2.
//
//   x_mid = x_in + a Wp^T + bp ; xn = LayerNorm2(x_mid) ; u = xn W1^T + b1 ; h = gelu(u) ; out = x_mid + h W2^T + b2
//
// Mapping: the M token rows are cut into 16-token tiles, a workgroup takes 8 or 9 consecutive tiles, each of its (up to)
// 9 waves carries ONE tile through the whole chain.  Every product is computed transposed (A = weight rows, B = token
// rows): an accumulator tile then holds [feature 16nt + 4g + r][token c], i.e. the token on the lane and the FEATURES in
// the registers -- exactly the layout in which acc_to_frag turns pairs of accumulator tiles into the B operand of the
// NEXT product, which contracts over those features (common.h).  So LayerNorm2's output feeds fc1 and gelu(fc1) feeds fc2
// straight from registers: no LDS image of the normalised panel, none of the hidden chunk (the round-1 kernel, retired in
// round 3, parked both: 108 KB, two barriers per 48-row epilogue pass, 63 barriers per workgroup).  The k order acc_to_frag
// produces inside a 32-feature chunk (t < 4 -> 4g + t, else 16 + 4g + t - 4) is baked into the packed weight copies ("phi"
// order), so the weight fragments are plain lane-linear 16-B reads.  Biases are the accumulators' initial values.
//
// LDS holds only weights, as 1-KB MFMA fragments filled by LDS-DMA (global_load_lds, 16 B per lane: one instruction moves
// one fragment, no staging registers): three 48-KB buffers; slab p holds fc1's rows of the 64-wide hidden chunk p (24
// fragments) and fc2's k chunk p - 1 (24 fragments) and is fetched TWO slabs ahead; one barrier per slab (14 per workgroup
// at HID = 768).  Inside a slab a wave runs two steps of a three-stage software pipeline over 32-wide sub-chunks t:
//     { fc1 of sub-chunk t + 1 (12 MFMAs)  ||  GELU epilogue of sub-chunk t  ||  fc2 of sub-chunk t - 1 (12 MFMAs) }
// the three are independent, so their instructions are interleaved in one scheduling region (an MFMA, its weight
// fragment read, a slice of the epilogue's VALU work): the matrix pipe runs under the VALU-bound GELU instead of all
// waves of a SIMD alternating between an MFMA phase and a VALU phase in lock step behind the slab barriers.
//
// What training keeps of the hidden layer is h = gelu(u) (fc2's weight gradient) and g' = gelu'(u) (the backward's
// du = (dy W2) * g'): the derivative costs one more FMA here (Phi(u) and exp(-u^2/2) are already at hand) and saves the
// backward kernel the whole erf evaluation.  Inference (SAVE = false) writes nothing of the hidden layer.
// Row statistics (LayerNorm2, and the next block's LayerNorm1 on the output) are in-lane sums over the 48 accumulator
// values of a token plus two v_permlane swaps.  Rows past M in the last tile are computed as copies of row M - 1 (clamped
// row index for loads AND stores: identical values written to the same address), so there is no guarded path.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#ifndef T2_RB
#define T2_RB 6
#endif
#ifndef T2_NV
#define T2_NV 8
#endif

namespace vitpe {

struct Tail2Args {
  const void* a;        // [M,192] merged-head attention output
  const void* xin;      // [M,192] block input (residual)
  const void* wp;       // attn.proj.weight packed (kchunk 192, natural order)
  const float* bp;
  const float* gamma;   // norm2
  const float* beta;
  const void* w1;       // mlp.fc1.weight packed (kchunk 192, phi order)
  const float* b1;
  const void* w2;       // mlp.fc2.weight packed (kchunk 32, phi order)
  const float* b2;
  void* xmid;           // [M,192] out
  float* mean2;         // [M] LayerNorm2 statistics of x_mid, out
  float* rstd2;
  void* xn_out;         // [M,192] LayerNorm2(x_mid), nullable
  void* gp_out;         // [M,HID] gelu'(u) as IEEE half (SAVE only)
  void* h_out;          // [M,HID] gelu(u)    (SAVE only)
  void* out;            // [M,192]
  float* mean_out;      // statistics of the output rows (next norm1), nullable (both or neither)
  float* rstd_out;
  int M, HID;
  float eps2, eps_next;
  unsigned long long* census;   // CENSUS instantiation only (include/vitpe_debug.h)
};

constexpr int T2_D = 192, T2_NT = 12, T2_KS = 6, T2_WAVES = 9;   // compute waves per workgroup
constexpr int T2_THREADS = 64 * (T2_WAVES + 2);                      // + two loader waves
constexpr int T2F_THREADS = 64 * (T2_WAVES + 3);                     // forward: + the helper wave of a split ninth tile
constexpr int T2_MAXHID = 1536, T2_NFLAG = T2_MAXHID / 64 + 10;   // slab flags (+ the fused qkv-gradient slabs of the backward)
constexpr int T2_CH = 64, T2_CNT = T2_CH / 16, T2_CKS = T2_CH / 32;   // hidden chunk per slab: output tiles of fc1, k steps of fc2
constexpr int T2_HALF = T2_CNT * T2_KS;                               // 24 fragments of fc1, 24 of fc2 (12 x 2) per slab
constexpr int T2_SLABF = 2 * T2_HALF;                                 // fragments per slab buffer
static_assert(T2_NT * T2_CKS == T2_HALF, "slab halves");

// lanes c, c+16, c+32, c+48 hold different features of the same token: sum across the four groups
VITPE_DEV float t2_xg_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Output rows are written with BUFFER stores: (descriptor of the workgroup's rows of the tensor, byte offset of the lane's
// row) instead of a pointer, so that the cache policy of every tensor is a compile-time choice (`aux`: 0 = write-back L2,
// 2 = nt, 16 = sc1: written through the L2 as it goes, nothing left for the end-of-kernel write-back).
#ifndef T2_LOADER_PRIO
#define T2_LOADER_PRIO 3  // s_setprio of the LDS-DMA loader waves: they share SIMDs 1 and 2 with two compute waves each, and a
                          // slab issued late is a slab every compute wave waits for (A/B at 0: step 1.523 -> 1.501 ms)
#endif
#ifndef T2_AUX_HID
#define T2_AUX_HID 2      // h, g', LayerNorm2 output, du: read again by a LATER kernel only (weight gradients / backward)
#endif
#ifndef T2_AUX_ROW
#define T2_AUX_ROW 0      // x_mid, out, da, dx: the next kernel's input
#endif
struct T2Row {
  __amdgpu_buffer_rsrc_t rsrc;
  int off;   // bytes
  VITPE_DEV T2Row operator+(int elems) const { return T2Row{rsrc, off + 2 * elems}; }
};
// rows [row0, ...) of a [*, ld] bf16 tensor; the lane's row `row` >= row0 (wave-uniform base, 32-bit offsets)
VITPE_DEV T2Row t2_row(void* base, int row0, int row, int ld) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(base) + (size_t)row0 * ld, 0, 0x7fffffff, 0x00020000);
  return T2Row{r, (row - row0) * ld * 2};
}
typedef __attribute__((ext_vector_type(4))) unsigned t2_u32x4;

// 16-B store of a tile pair: lane (c, g) holds features 16nt + 4g + r (r < 4) of token c for two adjacent tiles; one
// v_permlane16_swap per dword gives every lane 8 CONTIGUOUS features (see attn.hip)
template <int AUX>
VITPE_DEV void t2_store_pair(const T2Row& row, int nt0, int g, const f32x4& o0, const f32x4& o1) {
  uint32_t lo[2], hi[2];
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    bf16x2 pa, pb;
    pa[0] = (bf16)o0[2 * w2]; pa[1] = (bf16)o0[2 * w2 + 1];
    pb[0] = (bf16)o1[2 * w2]; pb[1] = (bf16)o1[2 * w2 + 1];
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, pa), __builtin_bit_cast(uint32_t, pb), false, false);
    lo[w2] = r[0]; hi[w2] = r[1];
  }
  const int f0 = 16 * (nt0 + (g & 1)) + 8 * (g >> 1);
  __builtin_amdgcn_raw_buffer_store_b128((t2_u32x4){lo[0], lo[1], hi[0], hi[1]}, row.rsrc, row.off + 2 * f0, 0, AUX);
}

// The same for gelu'(u), kept as IEEE half (round toward zero, v_cvt_pkrtz_f16_f32: one instruction per pair like the bf16
// conversion).  |gelu'| <= 1.13 needs no exponent range, and the backward multiplies EVERY du element by this value: rounded
// to bf16 (8 significant bits) it put 2^-9 of relative noise on all of them, which the cancelling sums of the backward
// (the shared polynomial-RPE coefficients: a sum over heads, layers and token pairs that mostly cancels) amplified to
// 9 % of their max norm; 11 bits bring that back to what recomputing gelu'(u) from the bf16 u gives (round-3 log).
template <int AUX>
VITPE_DEV void t2_store_pair_f16(const T2Row& row, int nt0, int g, const f32x4& o0, const f32x4& o1) {
  uint32_t lo[2], hi[2];
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const uint32_t pa = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o0[2 * w2], o0[2 * w2 + 1]));
    const uint32_t pb = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(o1[2 * w2], o1[2 * w2 + 1]));
    const auto r = __builtin_amdgcn_permlane16_swap(pa, pb, false, false);
    lo[w2] = r[0]; hi[w2] = r[1];
  }
  const int f0 = 16 * (nt0 + (g & 1)) + 8 * (g >> 1);
  __builtin_amdgcn_raw_buffer_store_b128((t2_u32x4){lo[0], lo[1], hi[0], hi[1]}, row.rsrc, row.off + 2 * f0, 0, AUX);
}

// The same 16-B row piece from the B FRAGMENT acc_to_frag made of a tile pair (lane (c, g): tile 2ks's features 4g .. 4g + 3
// in dwords 0 - 1, tile 2ks + 1's in dwords 2 - 3): LayerNorm2's output row, written from what fc1 consumes.
template <int AUX>
VITPE_DEV void t2_store_frag(const T2Row& row, int ks, int g, const Frag<bf16>& f) {
  const t2_u32x4 d = __builtin_bit_cast(t2_u32x4, f.v);
  const auto r0 = __builtin_amdgcn_permlane16_swap(d[0], d[2], false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(d[1], d[3], false, false);
  const int f0 = 16 * (2 * ks + (g & 1)) + 8 * (g >> 1);
  __builtin_amdgcn_raw_buffer_store_b128((t2_u32x4){r0[0], r1[0], r0[1], r1[1]}, row.rsrc, row.off + 2 * f0, 0, AUX);
}

// acc[nt] += sum_ks W(nt, ks) x bf[ks] over NTL x KSL weight fragments in LDS (fragment (nt, ks) at wb + (nt * KSL + ks) * 512
// elements, wb already lane-offset), k step outer / output tile inner so consecutive MFMAs hit different accumulators.
// Software pipelined by hand as a ring of R fragment registers: the ds_read of fragment q + R - 1 is issued right before
// the MFMA of fragment q, so every read has R - 1 MFMAs to land.  (Left alone the compiler reads one fragment ahead and
// waits lgkmcnt(1) before every MFMA.)  t2_gemm_code emits the instructions, t2_gemm_order pins their order with NV VALU
// instructions of whatever else is in the scheduling region after every MFMA.
template <int NTL, int KSL, int R>
VITPE_DEV void t2_gemm_code(const bf16* wb, const Frag<bf16>* bf, f32x4* acc) {
  constexpr int TOT = NTL * KSL;
  static_assert(TOT >= R, "ring deeper than the product");
  Frag<bf16> w[R];
#pragma unroll
  for (int i = 0; i < R - 1; ++i) w[i] = ld_frag(wb + ((i % NTL) * KSL + i / NTL) * 512);
#pragma unroll
  for (int q = 0; q < TOT; ++q) {
    if (q + R - 1 < TOT) {
      const int n = q + R - 1;
      w[n % R] = ld_frag(wb + ((n % NTL) * KSL + n / NTL) * 512);
    }
    mma(w[q % R], bf[q / NTL], acc[q % NTL]);
  }
}
template <int TOT, int R, int NV>
VITPE_DEV void t2_gemm_order() {
  __builtin_amdgcn_sched_group_barrier(0x100, R - 1, 0);
#pragma unroll
  for (int q = 0; q < TOT; ++q) {
    if (q + R - 1 < TOT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
    if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);           // NV VALU instructions
  }
}
template <int NTL, int KSL, int R>
VITPE_DEV void t2_gemm(const bf16* wb, const Frag<bf16>* bf, f32x4* acc) {
  __builtin_amdgcn_sched_barrier(0);
  t2_gemm_code<NTL, KSL, R>(wb, bf, acc);
  t2_gemm_order<NTL * KSL, R, 0>();
  __builtin_amdgcn_sched_barrier(0);
}

// h = gelu(x) = x Phi(x) and g = gelu'(x) = Phi(x) + x phi(x), exact-erf form (nn.GELU default, reference vit.py:111), in 15
// VALU + 2 transcendental instructions.  Same Abramowitz-Stegun 7.1.26 erfc as common.h's gelu_cdf (|err| <= 1.5e-7) with
// everything foldable folded:  T = 1 / (1 + p|x|/sqrt2);  E = phi(x) = exp2(-x^2 log2(e)/2 + log2(1/sqrt(2 pi)));
// H = erfc(|x|/sqrt2)/2 = E * T * poly(T) with the coefficients pre-divided by 2 phi(0);
// h = max(x, 0) - |x| H   (x >= 0: x (1 - H); x < 0: x H);   r = |x| E - H;   g = x >= 0 ? 1 + r : -r.
// Checked against float64 erf over [-8, 8]: |h - ref| <= 3.3e-7, |g - ref| <= 2.6e-7.
VITPE_DEV void t2_gelu(float x, float& h, float& g) {
  constexpr float C = 0.39894228040143267794f;               // phi(0) = 1 / sqrt(2 pi)
  const float ax = fabsf(x);
  const float T = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  float q = fmaf(T, 1.061405429f / (2.f * C), -1.453152027f / (2.f * C));
  q = fmaf(T, q, 1.421413741f / (2.f * C));
  q = fmaf(T, q, -0.284496736f / (2.f * C));
  q = fmaf(T, q, 0.254829592f / (2.f * C));
  const float E = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044448170368f, -1.32574806473615827f));
  const float H = (q * T) * E;
  h = fmaf(-ax, H, fmaxf(x, 0.f));
  const float r = fmaf(ax, E, -H);
  g = x >= 0.f ? 1.0f + r : -r;
}

// One step of the three-stage pipeline over 32-wide hidden sub-chunks t (forward; the backward kernel runs the same
// pipeline on the transposed weights with its own G stage):
//   F1: fc1 product of sub-chunk t + 1  (12 MFMAs: 2 output tiles x 6 k steps, B = the LayerNorm2 fragments)  -> a1n
//   G : elementwise epilogue of sub-chunk t, a callable (forward: GELU; backward: times gelu'(u))
//   F2: fc2 product of sub-chunk t - 1  (12 MFMAs: 12 output tiles x 1 k step, B = hprev)                      -> acc2
// The three are independent, so they go into ONE scheduling region: the two products' fragments alternate through one
// ring of R registers and every MFMA is followed by NV VALU instructions of the epilogue.
// w1f / w2f: lane-offset LDS pointers to the sub-chunk's fragments (fc1: (tile * 6 + k step) * 512, fc2: tile * 512).
template <bool F1, bool F2, int NV, int EXP, int R = 8, class GFn>
VITPE_DEV void t2_step(const bf16* w1f, const bf16* w2f, const Frag<bf16> (&bf)[T2_KS], f32x4 (&a1n)[2],
                       const Frag<bf16>& hprev, f32x4 (&acc2)[T2_NT], GFn gfn) {
  constexpr int TOT = (F1 ? 12 : 0) + (F2 ? 12 : 0);
  __builtin_amdgcn_sched_barrier(0);
  if (TOT > 0) {
    // unified fragment list: both products -> even q = fc2 fragment q / 2, odd q = fc1 fragment q / 2
    auto frag_ptr = [&](int q) -> const bf16* {
      const bool is2 = (F1 && F2) ? (q % 2 == 0) : F2;
      const int j = (F1 && F2) ? q / 2 : q;
      return is2 ? w2f + j * 512 : w1f + ((j % 2) * T2_KS + j / 2) * 512;     // fc1: k step outer, tile inner
    };
    Frag<bf16> w[R];
#pragma unroll
    for (int i = 0; i < R - 1; ++i) w[i] = ld_frag(frag_ptr((EXP & 1) ? (i & ~3) : i));
#pragma unroll
    for (int q = 0; q < TOT; ++q) {
      if (q + R - 1 < TOT) w[(q + R - 1) % R] = ld_frag(frag_ptr((EXP & 1) ? ((q + R - 1) & ~3) : q + R - 1));
      const bool is2 = (F1 && F2) ? (q % 2 == 0) : F2;
      const int j = (F1 && F2) ? q / 2 : q;
      if (is2) mma(w[q % R], hprev, acc2[j]);
      else mma(w[q % R], bf[j / 2], a1n[j % 2]);
    }
  }
  gfn();      // the G stage (nothing when NV == 0)
  if (TOT > 0) {
    __builtin_amdgcn_sched_group_barrier(0x100, R - 1, 0);
#pragma unroll
    for (int q = 0; q < TOT; ++q) {
      if (q + R - 1 < TOT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);           // NV VALU instructions of the G stage
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// forward: G = GELU of sub-chunk t (a1c = u with bias -> h fragment hnew; SAVE: h and g' rows stored)
template <bool F1, bool G, bool F2, bool SAVE, int EXP = 0>
VITPE_DEV void t2_substep(const bf16* w1f, const bf16* w2f, const Frag<bf16> (&bf)[T2_KS], f32x4 (&a1n)[2],
                          const f32x4 (&a1c)[2], const Frag<bf16>& hprev, Frag<bf16>& hnew, f32x4 (&acc2)[T2_NT],
                          const T2Row& gpr, const T2Row& hr, int g) {
  constexpr int NV = !G ? 0 : (F1 && F2) ? T2_NV : 2 * T2_NV;
  t2_step<F1, F2, NV, EXP>(w1f, w2f, bf, a1n, hprev, acc2, [&]() {
    if (!G) return;
    // scalar fp32 on purpose: packed f32 VALU (v_pk_mul/fma_f32) issues several times slower than two scalar
    // instructions beside MFMAs on gfx950 (MI355X_MICROARCH.md, issue-cost table), and this epilogue IS the bound
    f32x4 gp[2], hh[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = a1c[q][t];
        if (EXP & 2) { hh[q][t] = 0.5f * x; gp[q][t] = x; continue; }
        float hv, gv;
        t2_gelu(x, hv, gv);
        hh[q][t] = hv;
        gp[q][t] = gv;
      }
    }
    if (SAVE && !(EXP & 4)) {
      t2_store_pair_f16<T2_AUX_HID>(gpr, 0, g, gp[0], gp[1]);  // read again only in backward; IEEE half (see t2_store_pair_f16)
      t2_store_pair<T2_AUX_HID>(hr, 0, g, hh[0], hh[1]);
    }
    hnew = acc_to_frag<bf16>(hh[0], hh[1]);
  });
}

template <bool SAVE, bool CENSUS, int EXP = 0>
__global__ __launch_bounds__(T2F_THREADS) void block_tail2_fwd_kernel(Tail2Args a) {
  using T = bf16;
  constexpr int D = T2_D, NT = T2_NT, KS = T2_KS;
  __shared__ __attribute__((aligned(16))) T sW[3 * T2_SLABF * 512];
  __shared__ __attribute__((aligned(16))) float sPar[4 * T2_D + T2_MAXHID];   // bp | gamma | beta | b2 | b1
  __shared__ int sReady[T2_NFLAG], sDone[T2_NFLAG], sComb;   // loader <-> compute handshakes (see below)

  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HID = a.HID, nchunk = HID / T2_CH;
  // tiles of this workgroup / wave
  const int ntiles = (a.M + 15) / 16, base = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  const int tile0 = (int)blockIdx.x * base + min((int)blockIdx.x, rem), ntile_wg = base + ((int)blockIdx.x < rem ? 1 : 0);
  // A workgroup with NINE tiles would put three compute waves on SIMD 0 and finish 26 % after the eight-tile ones (the
  // kernel's duration).  Its ninth tile is therefore SPLIT between wave 8 (SIMD 0) and wave 11 (SIMD 3, which hosts only
  // two compute waves and no loader): both run the proj + LayerNorm2 prologue for the tile, wave 8 then takes the even
  // 64-wide hidden chunks and wave 11 the odd ones (fc1 + GELU of chunk p while slab p is resident, fc2 of chunk p under
  // slab p + 1), and wave 11's partial fc2 sums reach wave 8 through a slab buffer that has gone idle.
  const bool split = ntile_wg == T2_WAVES;
  const int half = !split ? -1 : wave == T2_WAVES - 1 ? 0 : wave == T2_WAVES + 2 ? 1 : -1;   // -1: a whole tile
  const bool active = wave < ntile_wg || half == 1;           // wave-uniform
  const int nsig = ntile_wg + (split ? 1 : 0);                // compute waves that count themselves into sDone[]
  const int mytile = tile0 + (half == 1 ? T2_WAVES - 1 : wave);
  const int rowc = min(16 * mytile + c, a.M - 1);            // rows past M: copies of row M - 1

  unsigned long long acc_sync = 0, acc_fc1 = 0, acc_mix = 0, tm0 = 0;
  auto now = [&]() -> unsigned long long { return CENSUS ? __builtin_amdgcn_s_memtime() : 0ull; };
  auto stamp = [&](int slot) {
    if (CENSUS && lane == 0 && wave < T2_WAVES) {
      a.census[((size_t)blockIdx.x * T2_WAVES + wave) * 16 + slot] = __builtin_amdgcn_s_memtime();
      // slots 13 / 14: the chip-wide 100-MHz clock at the first and the last stamp (s_memtime counts per XCD)
      if (slot == 0 || slot == 9) a.census[((size_t)blockIdx.x * T2_WAVES + wave) * 16 + (slot == 0 ? 13 : 14)] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
  auto dma1 = [&](const T* src_frag, int dst_frag) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_frag + lane * 8),
                                     (__attribute__((address_space(3))) void*)(sW + dst_frag * 512), 16, 0, 0);
  };
  // Wp -> fragments [0, 72): every wave issues its share (the first product waits for these)
#pragma unroll
  for (int i = 0; i < (NT * KS + T2F_THREADS / 64 - 1) / (T2F_THREADS / 64); ++i) {
    const int f = wave + (T2F_THREADS / 64) * i;
    if (f < NT * KS) dma1(reinterpret_cast<const T*>(a.wp) + (size_t)f * 512, f);
  }
  if (threadIdx.x < T2_NFLAG) { sReady[threadIdx.x] = 0; sDone[threadIdx.x] = 0; }
  if (threadIdx.x == 0) sComb = 0;
  for (int i = threadIdx.x; i < 4 * D + HID; i += T2F_THREADS) {     // parameters -> LDS
    const int k = i / D;
    sPar[i] = k == 0 ? a.bp[i] : k == 1 ? a.gamma[i - D] : k == 2 ? a.beta[i - 2 * D] : k == 3 ? a.b2[i - 3 * D] : a.b1[i - 4 * D];
  }

  // ---- waves 9 and 10: the loaders.  All further weight traffic of the workgroup is their LDS-DMA (an LDS-DMA piece costs
  // its issuing wave ~100 cycles of issue -- MI355X_MICROARCH.md, and measured here: one loader sustains a 48-piece slab
  // per 4.9 K cycles, which is why there are two; spread over the compute waves the pieces cost ~1 K cycles per wave and
  // slab right behind every barrier), and only THEY wait on vmcnt: the compute waves never wait for their own stores.
  // Wave 9 moves the fc1 halves of slabs 0 .. nchunk - 1, wave 10 the fc2 halves of slabs 1 .. nchunk, 24 pieces each.
  // After the one barrier that publishes Wp and the parameters there are NO workgroup barriers: a loader counts itself
  // into sReady[s] when its half of slab s has landed, a compute wave counts itself into sDone[] when it has consumed a
  // slab (sDone[0]: the proj fragments, sDone[s + 1]: slab s), and a loader refills a buffer when all active waves have
  // left it.  The compute waves never wait for each other, so the waves of a SIMD drift apart and one's MFMAs run under
  // another's GELU.
  if (wave == T2_WAVES || wave == T2_WAVES + 1) {
    __builtin_amdgcn_s_setprio(T2_LOADER_PRIO);
    const bool is1 = wave == T2_WAVES;                 // fc1 halves : fc2 halves
    const int first = is1 ? 0 : 1, last = is1 ? nchunk - 1 : nchunk;
    const T* const src = is1 ? reinterpret_cast<const T*>(a.w1) : reinterpret_cast<const T*>(a.w2) - (size_t)T2_HALF * 512;
    const int half = is1 ? 0 : T2_HALF;
    auto dma_half = [&](int sl) {                      // this loader's half of slab sl -> buffer (sl + 2) % 3
      const int b0 = ((sl + 2) % 3) * T2_SLABF + half;
#pragma unroll
      for (int f = 0; f < T2_HALF; ++f) dma1(src + ((size_t)sl * T2_HALF + f) * 512, b0 + f);
    };
    auto wait_done = [&](int k) {     // all active compute waves have counted themselves into sDone[k]
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sDone[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < nsig) __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
    };
    auto raise = [&](int sl) {
      asm volatile("" ::: "memory");
      if (lane == 0) atomicAdd(&sReady[sl], 1);
    };
    __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0): this wave's Wp pieces, LDS parameter stores
    asm volatile("s_barrier" ::: "memory");     // the one barrier
    if (is1) {
      dma_half(0);                        // -> buffer 2 (not under Wp)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      raise(0);
    }
    wait_done(0);                         // everyone past the proj product: its fragments may be overwritten
    dma_half(1);
    if (2 <= last) dma_half(2);
    for (int sl = 1; sl <= last; ++sl) {
      // slab sl must have landed; the only pieces issued after it are slab sl + 1's
      if (sl + 1 <= last) __builtin_amdgcn_s_waitcnt(0x4F78);   // vmcnt(24)
      else __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
      raise(sl);
      if (sl + 2 <= last) {
        wait_done(sl);                    // sDone[sl] counts slab sl - 1: its buffer is the one slab sl + 2 goes to
        dma_half(sl + 2);
      }
    }
    return;
  }
  auto wait_ready = [&](int sl) {       // both halves of slab sl (one for the first and the last slab) have landed
    const int need = (sl < nchunk ? 1 : 0) + (sl >= 1 ? 1 : 0);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sReady[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  auto signal_done = [&](int k) {       // this wave's LDS reads of the buffer are complete (consumed by MFMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&sDone[k], 1);
  };

  // The rest of the kernel exists twice -- whole-tile waves and the two half-tile waves of a nine-tile workgroup -- and is
  // entered through ONE wave-uniform branch right here: with the branch around the main loop only, the two paths' register
  // assignments met in the middle of the kernel and the compiler spilled 58 registers around it for every wave.
  auto body = [&](auto hm) {
    constexpr bool HALFW = decltype(hm)::value;
    // ---- compute waves: this wave's tokens as B fragments (natural k order), straight from global ---------------------------
    Frag<T> bf[KS];
    if (active) {
      const T* ar = reinterpret_cast<const T*>(a.a) + (size_t)rowc * D + 8 * g;
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) bf[ks] = ld_frag(ar + 32 * ks);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0) lgkmcnt(0): Wp pieces, token fragments, LDS parameter stores
    asm volatile("s_barrier" ::: "memory");       // the one barrier
    if (!active) return;
    stamp(1);
    // the residual rows: issued now (no LDS-DMA of this wave is in flight any more, so the compiler's own vmcnt waits are
    // counted ones again) and landing under the first product
    bf16x4 xres[NT];
    {
      const T* xr = reinterpret_cast<const T*>(a.xin) + (size_t)rowc * D + 4 * g;
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt) xres[nt] = *reinterpret_cast<const bf16x4*>(xr + 16 * nt);
    }

    // ---- x_mid = x_in + proj + bias; LayerNorm2 statistics; xn -> B fragments of fc1 --------------------------------------
    T* const xmr = reinterpret_cast<T*>(a.xmid) + (size_t)rowc * D;
    const T2Row xmrow = t2_row(a.xmid, 16 * tile0, rowc, D);
    const float invD = 1.0f / (float)D;
    {
      f32x4 acc[NT];
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = *reinterpret_cast<const f32x4*>(sPar + 16 * nt + 4 * g);
      t2_gemm<NT, KS, 12>(sW + lane * 8, bf, acc);
      signal_done(0);
      stamp(2);
      float s1 = 0.f;
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
  #pragma unroll
        for (int r = 0; r < 4; ++r) { acc[nt][r] = to_f32(from_f32<T>(acc[nt][r] + (float)xres[nt][r])); s1 += acc[nt][r]; }   // values as stored
      }
  #pragma unroll
      for (int nt = 0; nt < NT; nt += 2) t2_store_pair<T2_AUX_ROW>(xmrow, nt, g, acc[nt], acc[nt + 1]);
      __builtin_amdgcn_sched_barrier(0);   // (x_mid's conversions stay here: floated past the variance pass they keep 48 more values alive)
      const float mean = t2_xg_sum(s1) * invD;
      float s2 = 0.f;
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt)
  #pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc[nt][r] - mean; s2 += d * d; }
      const float rstd = 1.0f / sqrtf(t2_xg_sum(s2) * invD + a.eps2);
      if (g == 0) { a.mean2[rowc] = mean; a.rstd2[rowc] = rstd; }
      const T2Row xnr = t2_row(a.xn_out, 16 * tile0, rowc, D);
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
  #pragma unroll
        for (int nt = 2 * ks; nt < 2 * ks + 2; ++nt) {
          __builtin_amdgcn_sched_barrier(0);
          const f32x4 gv = *reinterpret_cast<const f32x4*>(sPar + D + 16 * nt + 4 * g);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sPar + 2 * D + 16 * nt + 4 * g);
  #pragma unroll
          for (int r = 0; r < 4; ++r) acc[nt][r] = fmaf((acc[nt][r] - mean) * rstd, gv[r], bv[r]);
        }
        bf[ks] = acc_to_frag<T>(acc[2 * ks], acc[2 * ks + 1]);   // phi order inside the chunk
        // (LayerNorm2's output row piece -- fc1's weight gradient reads it -- IS this fragment.  Writing the rows later, from
        //  the fragments, a few slabs into the main loop instead of in this burst behind x_mid's was tried: no change, 63 us
        //  either way; what the 12.8 MB cost shows only with HBM-resident operands, not in the waves' lifetimes: round-3 log)
        if (a.xn_out != nullptr && (!HALFW || half == 0)) t2_store_frag<T2_AUX_HID>(xnr, ks, g, bf[ks]);
      }
    }
    // ---- the MLP branch: slab p = fc1 rows of the 64-wide chunk p | fc2 k chunk p - 1, two pipeline steps per slab ----------
    f32x4 acc2[NT];
  #pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc2[nt] = *reinterpret_cast<const f32x4*>(sPar + 3 * D + 16 * nt + 4 * g);   // b2
    const T2Row gpr = t2_row(a.gp_out, 16 * tile0, rowc, HID), hr = t2_row(a.h_out, 16 * tile0, rowc, HID);   // (unused unless SAVE)
    f32x4 aX[2], aY[2];          // fc1 accumulators of the even / odd 32-wide sub-chunk in flight
    Frag<T> hP, hQ;              // gelu fragments of the even / odd sub-chunk in flight
    const float* const b1l = sPar + 4 * D + 4 * g;
    auto bias1 = [&](f32x4 (&acc1)[2], int t) {     // fc1 bias of sub-chunk t = the accumulators' initial value
      acc1[0] = *reinterpret_cast<const f32x4*>(b1l + 32 * t);
      acc1[1] = *reinterpret_cast<const f32x4*>(b1l + 32 * t + 16);
    };

    if constexpr (HALFW) {
      // ---- half of a split tile: chunk p's fc1 + GELU while slab p is resident, its fc2 under slab p + 1 (see the top) ---------
      if (half == 1) {
  #pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};       // the bias is wave 8's
      }
      for (int p = 0; p <= nchunk; ++p) {
        wait_ready(p);
        const T* wb = sW + ((p + 2) % 3) * T2_SLABF * 512 + lane * 8;
        if (p < nchunk && (p & 1) == half) {
          bias1(aX, 2 * p);
          t2_substep<true, false, false, SAVE, EXP>(wb, wb, bf, aX, aY, hQ, hQ, acc2, gpr, hr, g);
          bias1(aY, 2 * p + 1);
          t2_substep<true, true, false, SAVE, EXP>(wb + 2 * KS * 512, wb, bf, aY, aX, hQ, hP, acc2, gpr + 32 * (2 * p),
                                                   hr + 32 * (2 * p), g);
          t2_substep<false, true, false, SAVE, EXP>(wb, wb, bf, aX, aY, hQ, hQ, acc2, gpr + 32 * (2 * p + 1), hr + 32 * (2 * p + 1), g);
        } else if (p >= 1 && ((p - 1) & 1) == half) {
          const T* w2b = wb + T2_HALF * 512;
          Frag<T> hdummy;
          t2_substep<false, false, true, SAVE, EXP>(w2b, w2b, bf, aX, aY, hP, hdummy, acc2, gpr, hr, g);
          t2_substep<false, false, true, SAVE, EXP>(w2b, w2b + NT * 512, bf, aX, aY, hQ, hdummy, acc2, gpr, hr, g);
        }
        signal_done(p + 1);
      }
      // wave 11's partial sums -> wave 8 through the buffer slab nchunk - 2 lived in (idle once everyone has consumed it)
      float* const comb = reinterpret_cast<float*>(sW + (nchunk % 3) * T2_SLABF * 512) + lane;
      if (half == 1) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sDone[nchunk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < nsig)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
  #pragma unroll
        for (int nt = 0; nt < NT; ++nt)
  #pragma unroll
          for (int r = 0; r < 4; ++r) comb[(nt * 4 + r) * 64] = acc2[nt][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&sComb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sComb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0)
        __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt)
  #pragma unroll
        for (int r = 0; r < 4; ++r) acc2[nt][r] += comb[(nt * 4 + r) * 64];
    } else {
      stamp(3);
      wait_ready(0);
      stamp(4);
      {
        const T* wb = sW + 2 * T2_SLABF * 512 + lane * 8;
        bias1(aX, 0);
        t2_substep<true, false, false, SAVE, EXP>(wb, wb, bf, aX, aY, hQ, hQ, acc2, gpr, hr, g);                           // F1_0
        bias1(aY, 1);
        t2_substep<true, true, false, SAVE, EXP>(wb + 2 * KS * 512, wb, bf, aY, aX, hQ, hP, acc2, gpr, hr, g);             // F1_1 G_0
        signal_done(1);
      }
      stamp(5);
      for (int p = 1; p < nchunk; ++p) {
        if (CENSUS) tm0 = now();
        wait_ready(p);
        if (CENSUS) { const unsigned long long t = now(); acc_sync += t - tm0; tm0 = t; }
        {
          const T* wb = sW + ((p + 2) % 3) * T2_SLABF * 512 + lane * 8;
          const T* w2b = wb + T2_HALF * 512;
          bias1(aX, 2 * p);
          t2_substep<true, true, true, SAVE, EXP>(wb, w2b, bf, aX, aY, hP, hQ, acc2, gpr + 32 * (2 * p - 1), hr + 32 * (2 * p - 1), g);
          if (CENSUS) { const unsigned long long t = now(); acc_fc1 += t - tm0; tm0 = t; }
          bias1(aY, 2 * p + 1);
          t2_substep<true, true, true, SAVE, EXP>(wb + 2 * KS * 512, w2b + NT * 512, bf, aY, aX, hQ, hP, acc2, gpr + 32 * (2 * p),
                                             hr + 32 * (2 * p), g);
          signal_done(p + 1);
          if (CENSUS) acc_mix += now() - tm0;
        }
      }
      stamp(6);
      wait_ready(nchunk);
      stamp(7);
      if (CENSUS && lane == 0) {
        unsigned long long* cz = a.census + ((size_t)blockIdx.x * T2_WAVES + wave) * 16;
        cz[10] = acc_sync; cz[11] = acc_fc1; cz[12] = acc_mix;
      }
      {
        const T* w2b = sW + ((nchunk + 2) % 3) * T2_SLABF * 512 + T2_HALF * 512 + lane * 8;
        const int tl = 2 * nchunk - 1;
        t2_substep<false, true, true, SAVE, EXP>(w2b, w2b, bf, aX, aY, hP, hQ, acc2, gpr + 32 * tl, hr + 32 * tl, g);        // G_last F2
        t2_substep<false, false, true, SAVE, EXP>(w2b, w2b + NT * 512, bf, aX, aY, hQ, hQ, acc2, gpr, hr, g);               // F2_last
      }
    }
    stamp(8);

    // ---- out = x_mid + fc2 + bias; statistics for the next block's LayerNorm1 ----------------------------------------------
    {
      const T2Row outr = t2_row(a.out, 16 * tile0, rowc, D);
      float s1 = 0.f;
  #pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 rv = ld4(xmr + 16 * nt + 4 * g);     // (this lane's own earlier stores: same wave, same addresses)
  #pragma unroll
        for (int r = 0; r < 4; ++r) { acc2[nt][r] = to_f32(from_f32<T>(acc2[nt][r] + rv[r])); s1 += acc2[nt][r]; }
      }
  #pragma unroll
      for (int nt = 0; nt < NT; nt += 2) t2_store_pair<T2_AUX_ROW>(outr, nt, g, acc2[nt], acc2[nt + 1]);
      if (a.mean_out != nullptr) {
        const float mean = t2_xg_sum(s1) * invD;
        float s2 = 0.f;
  #pragma unroll
        for (int nt = 0; nt < NT; ++nt)
  #pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = acc2[nt][r] - mean; s2 += d * d; }
        const float var = t2_xg_sum(s2) * invD;
        if (g == 0) { a.mean_out[rowc] = mean; a.rstd_out[rowc] = 1.0f / sqrtf(var + a.eps_next); }
      }
    }
    stamp(9);
  };
  if (half >= 0) body(std::true_type{});
  else body(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the block tail w.r.t. its inputs (the weight gradients are the grouped wgrad kernel's): the same pipeline on
// the transposed weights.  Per 16-token tile, everything transposed ([feature][token] accumulators):
//   dh  = dy W2            F1 stage: A = fc2.weight^T rows of the hidden sub-chunk (acc_to_frag k order), B = dy fragments
//   du  = dh * gelu'(u)    G stage: gp rows loaded 2 sub-chunks ahead, du rows stored (fc1's weight gradient reads them)
//   dxn = du W1            F2 stage: A = fc1.weight^T (k = hidden sub-chunk, acc_to_frag order), B = du fragments
//   dx_mid = dy + LayerNorm2'(dxn)   (row statistics saved by the forward; dgamma / dbeta: DPP row sums over the 16 tokens,
//                                     LDS atomics per workgroup, one global atomic per column and workgroup)
//   da  = dx_mid Wp        A = attn.proj.weight^T (acc_to_frag order), loaded into the slab buffers once they are free
struct Tail2BwdArgs {
  const void* dy;       // [M,192] gradient of the block output
  const void* gp;       // [M,HID] gelu'(u) as IEEE half (vitpe_block_tail2_fwd)
  const void* xmid;     // [M,192] LayerNorm2's input rows
  const float* mean2;   // [M]
  const float* rstd2;
  const float* gamma;   // norm2.weight [192]
  const void* w2t;      // pack(fc2.weight^T [HID,192], kchunk 192, phi)
  const void* w1t;      // pack(fc1.weight^T [192,HID], kchunk 32, phi)
  const void* wpt;      // pack(attn.proj.weight^T [192,192], kchunk 192, phi)
  void* du;             // [M,HID] out
  void* dxmid;          // [M,192] out
  void* da;             // [M,192] out
  float* dgamma;        // [192] accumulated
  float* dbeta;
  int M, HID;
  // PRE instantiation: the block ABOVE's qkv data gradient + LayerNorm1 backward + residual run first and PRODUCE dy
  // (dy is then written by this kernel, for the weight gradients and its own LayerNorm2 residual, not read at the start)
  const void* dqkv;     // [M,K1] gradient of that block's qkv projection
  const void* wqt;      // pack(qkv.weight^T [192,K1], kchunk 64, natural)
  const void* x1;       // [M,192] that block's input rows (LayerNorm1 input)
  const float* mean1;
  const float* rstd1;
  const float* gamma1;
  const void* dres1;    // [M,192] that block's d x_mid (residual)
  float* dgamma1;
  float* dbeta1;
  int K1;
};

// inverse of t2_store_pair: lane (c, g) loads the 8 contiguous features it would have stored and gets back the two tiles'
// [feature 4g + r][token c] values (v_permlane16_swap is an involution)
VITPE_DEV void t2_unpack_pair(const Chunk16& v, f32x4& o0, f32x4& o1) {
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const auto r = __builtin_amdgcn_permlane16_swap(v[w2], v[2 + w2], false, false);
    o0[2 * w2] = __uint_as_float(r[0] << 16);
    o0[2 * w2 + 1] = __uint_as_float(r[0] & 0xffff0000u);
    o1[2 * w2] = __uint_as_float(r[1] << 16);
    o1[2 * w2 + 1] = __uint_as_float(r[1] & 0xffff0000u);
  }
}

// ... of a row piece stored by t2_store_pair_f16 (gelu'(u), IEEE half)
VITPE_DEV void t2_unpack_pair_f16(const Chunk16& v, f32x4& o0, f32x4& o1) {
#pragma unroll
  for (int w2 = 0; w2 < 2; ++w2) {
    const auto r = __builtin_amdgcn_permlane16_swap(v[w2], v[2 + w2], false, false);
    // (scalar halves on purpose: with a 2 x half vector cast of the two results hipcc 7.2 multiplied BOTH tiles by the
    //  first result's values -- the second result of the swap was dead in the emitted code)
    const uint32_t ra = r[0], rb = r[1];
    o0[2 * w2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(ra & 0xffffu));
    o0[2 * w2 + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(ra >> 16));
    o1[2 * w2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rb & 0xffffu));
    o1[2 * w2 + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(rb >> 16));
  }
}

// Column sums of 8 values over the 16 lanes of a DPP row (the 16 tokens of a tile) as a halving butterfly: each step a
// lane keeps half of its values and adds its partner's copy of that half (partners: lane ^ 8, 7 - lane within the 8, ^ 2,
// ^ 1), 15 DPP moves instead of 32.  Lane c ends with the total of value 4 b3 + 2 b2 + b1 (bits of c; b0 ignored).
template <int CTRL> VITPE_DEV float t2_dppc(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
VITPE_DEV float t2_colsum8(const float (&t)[8], int c) {
  const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0, b1 = (c & 2) != 0;
  float u[4], w[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) u[k] = (b3 ? t[4 + k] : t[k]) + t2_dppc<0x128>(b3 ? t[k] : t[4 + k]);        // row_ror:8
#pragma unroll
  for (int k = 0; k < 2; ++k) w[k] = (b2 ? u[2 + k] : u[k]) + t2_dppc<0x141>(b2 ? u[k] : u[2 + k]);        // row_half_mirror
  const float x = (b1 ? w[1] : w[0]) + t2_dppc<0x4E>(b1 ? w[0] : w[1]);                                    // quad_perm [2,3,0,1]
  return x + t2_dppc<0xB1>(x);                                                                               // quad_perm [1,0,3,2]
}

// LayerNorm backward + residual on a transposed 16-token tile: acc = gradient w.r.t. the LayerNorm's OUTPUT on entry,
//   acc <- res + rstd (gy - mean_f(gy) - xhat mean_f(gy xhat)),  gy = acc gamma,  xhat = (x - mean) rstd
// rounded to the bf16 values stored to `orow`; the tile's dgamma / dbeta column sums (rows past M masked by `valid`) go to
// the workgroup's LDS accumulators sAcc[0..191 | 192..383].  xrow / rrow: this lane's LayerNorm-input / residual row + 4g.
VITPE_DEV void t2_ln_backward(f32x4 (&acc)[T2_NT], const bf16* xrow, const bf16* rrow, float mean, float rstd,
                              const float* sGam, float* sAcc, int c, int g, float valid, const T2Row& orow) {
  constexpr int D = T2_D, NT = T2_NT;
  bf16x4 xmv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) xmv[nt] = *reinterpret_cast<const bf16x4*>(xrow + 16 * nt);
  const float invD = 1.0f / (float)D;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 gam = *reinterpret_cast<const f32x4*>(sGam + 16 * nt + 4 * g);
    float tot[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xhat = ((float)xmv[nt][r] - mean) * rstd;
      const float dxn = acc[nt][r];
      tot[r] = dxn * xhat * valid;       // dgamma / dbeta contributions (rows past M masked)
      tot[4 + r] = dxn * valid;
      const float gy = dxn * gam[r];
      s1 += gy;
      s2 = fmaf(gy, xhat, s2);
    }
    // column sums over the tile's 16 tokens: even lane c of every row ends with total (c >> 1): one LDS atomic per tile
    const float sel = t2_colsum8(tot, c);
    // (the conditional atomic ends a basic block; without this pin the compiler sinks the whole s1 / s2 chains past all
    //  twelve blocks and keeps the 48 xhat values alive for them: 36 spilled registers)
    asm volatile("" : "+v"(s1), "+v"(s2));
    if (!(c & 1)) atomicAdd(&sAcc[(c < 8 ? 0 : D - 4) + 16 * nt + 4 * g + (c >> 1)], sel);
  }
  const float m1 = t2_xg_sum(s1) * invD, m2 = t2_xg_sum(s2) * invD;
  bf16x4 rv[NT];                                   // the residual rows
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) rv[nt] = *reinterpret_cast<const bf16x4*>(rrow + 16 * nt);
  // second pass: gy and xhat are recomputed from the packed rows (keeping 48 fp32 xhat values alive across the row sums
  // is what spilled); the empty asm stops the compiler from re-using the first pass's conversions
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    __builtin_amdgcn_sched_barrier(0);
    uint2 xw = __builtin_bit_cast(uint2, xmv[nt]);
    asm volatile("" : "+v"(xw.x), "+v"(xw.y));
    const bf16x4 xb = __builtin_bit_cast(bf16x4, xw);
    const f32x4 gam = *reinterpret_cast<const f32x4*>(sGam + 16 * nt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xhat = ((float)xb[r] - mean) * rstd;
      const float gy = acc[nt][r] * gam[r];
      acc[nt][r] = to_f32(from_f32<bf16>(fmaf(rstd, gy - m1 - xhat * m2, (float)rv[nt][r])));   // as stored
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; nt += 2) t2_store_pair<T2_AUX_ROW>(orow, nt, g, acc[nt], acc[nt + 1]);
}

// the last of the workgroup's `nactive` compute waves to arrive hands the LDS column sums to the global accumulators
VITPE_DEV void t2_flush_colsums(float* sAcc, int* sFin, int nactive, int lane, float* dgamma, float* dbeta) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int fin = 0;
  if (lane == 0) fin = atomicAdd(sFin, 1);
  fin = __builtin_amdgcn_readfirstlane(fin);
  if (fin == nactive - 1) {
    for (int i = lane; i < T2_D; i += 64) {
      atomicAdd(dgamma + i, __hip_atomic_load(&sAcc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      atomicAdd(dbeta + i, __hip_atomic_load(&sAcc[T2_D + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
  }
}

template <bool PRE>
__global__ __launch_bounds__(T2_THREADS) void block_tail2_bwd_kernel(Tail2BwdArgs a) {
  using T = bf16;
  constexpr int D = T2_D, NT = T2_NT, KS = T2_KS;
  __shared__ __attribute__((aligned(16))) T sW[3 * T2_SLABF * 512];
  __shared__ __attribute__((aligned(16))) float sGam[T2_D], sGam1[PRE ? T2_D : 4];
  __shared__ float sAcc[2 * T2_D], sAcc1[PRE ? 2 * T2_D : 4];     // dgamma | dbeta of this workgroup (norm2; PRE: + the upper norm1)
  __shared__ int sReady[T2_NFLAG], sDone[T2_NFLAG], sFin;

  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HID = a.HID, nchunk = HID / T2_CH, nsub = 2 * nchunk;
  // PRE: the qkv^T weight comes first, as Q slabs of two 64-deep k chunks (24 fragments each; the last slab may hold one)
  const int nkc = PRE ? a.K1 / 64 : 0, Q = (nkc + 1) / 2, QB = nkc / 2;
  const int ntiles = (a.M + 15) / 16, base = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  const int tile0 = (int)blockIdx.x * base + min((int)blockIdx.x, rem), ntile_wg = base + ((int)blockIdx.x < rem ? 1 : 0);
  const bool active = wave < ntile_wg;
  const int row = 16 * (tile0 + wave) + c;
  const int rowc = min(row, a.M - 1);                 // rows past M: computed as copies of row M - 1, masked out of the sums
  const float valid = row < a.M ? 1.0f : 0.0f;

  if (threadIdx.x < T2_NFLAG) { sReady[threadIdx.x] = 0; sDone[threadIdx.x] = 0; }
  if (threadIdx.x == 0) sFin = 0;
  for (int i = threadIdx.x; i < 2 * D; i += T2_THREADS) { sAcc[i] = 0.f; if (PRE) sAcc1[i] = 0.f; }
  for (int i = threadIdx.x; i < D; i += T2_THREADS) { sGam[i] = a.gamma[i]; if (PRE) sGam1[i] = a.gamma1[i]; }

  auto dma1 = [&](const T* src_frag, int dst_frag) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_frag + lane * 8),
                                     (__attribute__((address_space(3))) void*)(sW + dst_frag * 512), 16, 0, 0);
  };
  // ---- waves 9 and 10: the loaders (see the forward kernel).  One sequence of slabs g = 0 .. Q + nchunk: the Q qkv^T slabs
  // (PRE), then MLP slab s at g = Q + s (F1 half: fc2.weight^T rows of chunk s, F2 half: fc1.weight^T k chunk s - 1); slab g
  // lives in buffer (g + 2) % 3, wave 9 moves the first 24 fragments of a slab, wave 10 the second 24 (where they exist).
  // A slab may be issued once slab g - 3 has been consumed by every active wave (sDone[g - 2]).  When everything has
  // been consumed both load their half of attn.proj.weight^T into fragments [0, 72).
  if (wave >= T2_WAVES) {
    __builtin_amdgcn_s_setprio(T2_LOADER_PRIO);
    const bool is1 = wave == T2_WAVES;
    const int nq = is1 ? Q : QB, njobs = nq + nchunk;
    const int half = is1 ? 0 : T2_HALF;
    auto job_slab = [&](int j) { return j < nq ? j : Q + (j - nq) + (is1 ? 0 : 1); };
    auto dma_job = [&](int j) {
      const T* src;
      if (j < nq) src = reinterpret_cast<const T*>(a.wqt) + ((size_t)j * T2_SLABF + half) * 512;
      else src = (is1 ? reinterpret_cast<const T*>(a.w2t) : reinterpret_cast<const T*>(a.w1t)) + (size_t)(j - nq) * T2_HALF * 512;
      const int b0 = ((job_slab(j) + 2) % 3) * T2_SLABF + half;
#pragma unroll
      for (int f = 0; f < T2_HALF; ++f) dma1(src + (size_t)f * 512, b0 + f);
    };
    auto wait_done = [&](int k) {
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sDone[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < ntile_wg)
        __builtin_amdgcn_s_sleep(2);
      asm volatile("" ::: "memory");
    };
    auto raise = [&](int sl) {
      asm volatile("" ::: "memory");
      if (lane == 0) atomicAdd(&sReady[sl], 1);
    };
    __builtin_amdgcn_s_waitcnt(0x0070);         // lgkmcnt(0): the LDS initialisation above
    asm volatile("s_barrier" ::: "memory");     // the one barrier
    // Two jobs in flight.  A job's buffer is free when slab g - 3 has been consumed (sDone[g - 2]) -- but this loader must have
    // RAISED its own older slabs before it waits for them to be consumed (with the qkv slabs in front, wave 10's slab list
    // jumps from 3 to Q + 1: waiting for slab 3 to be consumed before raising slab 3 is a deadlock).
    int issued = 0, raised = 0;
    auto top_up = [&]() {
      while (issued < njobs && issued - raised < 2) {
        const int g2 = job_slab(issued);
        if (g2 >= 3) {
          if (raised < issued && job_slab(raised) <= g2 - 3) break;        // raise that one first
          wait_done(g2 - 2);
        }
        dma_job(issued++);
      }
    };
    top_up();
    while (raised < njobs) {
      if (issued - raised - 1 >= 1) __builtin_amdgcn_s_waitcnt(0x4F78);   // vmcnt(24): one younger job may be in flight
      else __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
      raise(job_slab(raised));
      ++raised;
      top_up();
    }
    wait_done(Q + nchunk + 1);                                 // every slab consumed by every active wave
#pragma unroll
    for (int f = 0; f < NT * KS / 2; ++f) {
      const int ff = (is1 ? 0 : NT * KS / 2) + f;
      dma1(reinterpret_cast<const T*>(a.wpt) + (size_t)ff * 512, ff);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    raise(Q + nchunk + 1);
    return;
  }
  auto wait_ready = [&](int gs) {      // both halves (where they exist) of global slab gs have landed
    const int s = gs - Q;
    const int need = gs < Q ? 1 + (2 * gs + 1 < nkc ? 1 : 0) : s > nchunk ? 2 : (s < nchunk ? 1 : 0) + (s >= 1 ? 1 : 0);
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sReady[gs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  auto signal_done = [&](int k) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&sDone[k], 1);
  };
  auto slab_ptr = [&](int gs) { return sW + ((gs + 2) % 3) * T2_SLABF * 512 + lane * 8; };

  // ---- compute waves -------------------------------------------------------------------------------------------------------
  Frag<T> bf[KS];
  const T* const dqr = PRE ? reinterpret_cast<const T*>(a.dqkv) + (size_t)rowc * a.K1 + 8 * g : nullptr;
  Frag<T> bqa[4], bqb[4];       // PRE: this wave's d_qkv rows as B fragments (natural k order), four 32-deep steps per slab
  if (active) {
    if (PRE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bqa[i] = ld_frag(dqr + 32 * min(i, a.K1 / 32 - 1));
    } else {
      // dy rows as B fragments in the acc_to_frag k order (what the PRE path produces from its accumulators)
      const T* dr = reinterpret_cast<const T*>(a.dy) + (size_t)rowc * D + 4 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(dr + 32 * ks), hi = *reinterpret_cast<const bf16x4*>(dr + 32 * ks + 16);
        bf[ks].v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0) lgkmcnt(0)
  asm volatile("s_barrier" ::: "memory");       // the one barrier
  if (!active) return;

  if (PRE) {
    // ---- the upper block's  dy = d x_mid' + LayerNorm1'(d_qkv Wqkv)  (vitpe_linear_lnbwd2's work), kept in registers -----------
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = a.K1 / 32;
    auto qslab = [&](int q, const Frag<T> (&cur)[4], Frag<T> (&nxt)[4]) {
      if (q + 1 < Q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nxt[i] = ld_frag(dqr + 32 * min(4 * (q + 1) + i, nks - 1));
      }
      wait_ready(q);
      const T* wb = slab_ptr(q);
      t2_gemm<NT, 2, 8>(wb, cur, acc);
      if (4 * q + 2 < nks) t2_gemm<NT, 2, 8>(wb + T2_HALF * 512, cur + 2, acc);
      signal_done(q + 1);
    };
    for (int q = 0; q < Q; q += 2) {
      qslab(q, bqa, bqb);
      if (q + 1 < Q) qslab(q + 1, bqb, bqa);
    }
    t2_ln_backward(acc, reinterpret_cast<const T*>(a.x1) + (size_t)rowc * D + 4 * g,
                   reinterpret_cast<const T*>(a.dres1) + (size_t)rowc * D + 4 * g, a.mean1[rowc], a.rstd1[rowc], sGam1, sAcc1,
                   c, g, valid, t2_row(const_cast<void*>(a.dy), 16 * tile0, rowc, D));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bf[ks] = acc_to_frag<T>(acc[2 * ks], acc[2 * ks + 1]);
  }

  const T* const gpr = reinterpret_cast<const T*>(a.gp) + (size_t)rowc * HID + 16 * (g & 1) + 8 * (g >> 1);
  const T2Row dur = t2_row(a.du, 16 * tile0, rowc, HID);
  Chunk16 gq[2];                                 // gelu'(u) rows of sub-chunks t, t + 1 in flight (slot t & 1)
#pragma unroll
  for (int t = 0; t < 2; ++t) gq[t] = *reinterpret_cast<const Chunk16*>(gpr + 32 * t);
  f32x4 acc2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 aX[2], aY[2];
  Frag<T> hP, hQ;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // G stage of sub-chunk t: du = dh * gelu'(u) -> stored, -> B fragment of the F2 product; refills its prefetch slot
  auto gstage = [&](const f32x4 (&a1c)[2], Chunk16& slot, int t, Frag<T>& hnew) {
    f32x4 g0, g1, d0, d1;
    t2_unpack_pair_f16(slot, g0, g1);
    if (t + 2 < nsub) slot = *reinterpret_cast<const Chunk16*>(gpr + 32 * (t + 2));
#pragma unroll
    for (int r = 0; r < 4; ++r) { d0[r] = a1c[0][r] * g0[r]; d1[r] = a1c[1][r] * g1[r]; }
    t2_store_pair<T2_AUX_HID>(dur + 32 * t, 0, g, d0, d1);
    hnew = acc_to_frag<T>(d0, d1);
  };
  constexpr int NVB = 3, RB = T2_RB;    // VALU instructions of the G stage per MFMA (it is short: the step is matrix-pipe bound)

  wait_ready(Q);
  {
    const T* wb = slab_ptr(Q);
    aX[0] = z4; aX[1] = z4;
    t2_step<true, false, 0, 0, RB>(wb, wb, bf, aX, hQ, acc2, [&]() {});                                             // F1_0
    aY[0] = z4; aY[1] = z4;
    t2_step<true, false, 2 * NVB, 0, RB>(wb + 2 * KS * 512, wb, bf, aY, hQ, acc2, [&]() { gstage(aX, gq[0], 0, hP); });   // F1_1 G_0
    signal_done(Q + 1);
  }
  for (int p = 1; p < nchunk; ++p) {
    wait_ready(Q + p);
    const T* wb = slab_ptr(Q + p);
    const T* w2b = wb + T2_HALF * 512;
    aX[0] = z4; aX[1] = z4;
    t2_step<true, true, NVB, 0, RB>(wb, w2b, bf, aX, hP, acc2, [&]() { gstage(aY, gq[1], 2 * p - 1, hQ); });
    aY[0] = z4; aY[1] = z4;
    t2_step<true, true, NVB, 0, RB>(wb + 2 * KS * 512, w2b + NT * 512, bf, aY, hQ, acc2, [&]() { gstage(aX, gq[0], 2 * p, hP); });
    signal_done(Q + p + 1);
  }
  wait_ready(Q + nchunk);
  {
    const T* w2b = slab_ptr(Q + nchunk) + T2_HALF * 512;
    const int tl = nsub - 1;
    t2_step<false, true, 2 * NVB, 0, RB>(w2b, w2b, bf, aX, hP, acc2, [&]() { gstage(aY, gq[1], tl, hQ); });
    t2_step<false, true, 0, 0, RB>(w2b, w2b + NT * 512, bf, aX, hQ, acc2, [&]() {});                                  // F2_last
    signal_done(Q + nchunk + 1);
  }
  // ---- LayerNorm2 backward + residual: dx_mid = dy + LayerNorm2'(dxn) ------------------------------------------------------------
  t2_ln_backward(acc2, reinterpret_cast<const T*>(a.xmid) + (size_t)rowc * D + 4 * g,
                 reinterpret_cast<const T*>(a.dy) + (size_t)rowc * D + 4 * g, a.mean2[rowc], a.rstd2[rowc], sGam, sAcc, c, g,
                 valid, t2_row(a.dxmid, 16 * tile0, rowc, D));
  // ---- da = dx_mid Wp ----------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bf[ks] = acc_to_frag<T>(acc2[2 * ks], acc2[2 * ks + 1]);
  f32x4 accA[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) accA[nt] = z4;
  wait_ready(Q + nchunk + 1);
  t2_gemm<NT, KS, 12>(sW + lane * 8, bf, accA);
  {
    const T2Row dar = t2_row(a.da, 16 * tile0, rowc, D);
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) t2_store_pair<T2_AUX_ROW>(dar, nt, g, accA[nt], accA[nt + 1]);
  }
  // ---- the last wave of the workgroup hands the column sums to the global accumulators ----------------------------------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int fin = 0;
  if (lane == 0) fin = atomicAdd(&sFin, 1);
  fin = __builtin_amdgcn_readfirstlane(fin);
  if (fin == ntile_wg - 1) {
    for (int i = lane; i < D; i += 64) {
      atomicAdd(a.dgamma + i, __hip_atomic_load(&sAcc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      atomicAdd(a.dbeta + i, __hip_atomic_load(&sAcc[D + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (PRE) {
        atomicAdd(a.dgamma1 + i, __hip_atomic_load(&sAcc1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        atomicAdd(a.dbeta1 + i, __hip_atomic_load(&sAcc1[D + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dx = dres + LayerNorm'(dY Wt^T): the data gradient of a Linear whose input is a LayerNorm output, that LayerNorm's
// backward, and the residual add (attn.qkv: dY = d_qkv [M,576], LayerNorm1, dres = d x_mid) -- the wave-per-tile mapping of
// the kernels above.  Wt packed = pack(weight^T [192, K], kchunk 64, natural): K / 192 slabs of 72 fragments, two LDS
// buffers, the two loader waves each move half a slab; a compute wave loads its 16 dY rows as B fragments straight from
// global, one 192-wide k chunk ahead of the product.
struct LnBwd2Args {
  const void* dy;       // [M,K]
  const void* wt;       // pack(W^T [192,K], 64, natural)
  const void* x;        // [M,192] LayerNorm input rows
  const float* mean;
  const float* rstd;
  const float* gamma;
  const void* dres;     // [M,192] residual gradient
  void* dx;             // [M,192] out
  float* dgamma;
  float* dbeta;
  int M, K;
};

__global__ __launch_bounds__(T2_THREADS) void ln_bwd2_kernel(LnBwd2Args a) {
  using T = bf16;
  constexpr int D = T2_D, NT = T2_NT, KS = T2_KS, SLF = NT * KS;    // 72 fragments per slab
  __shared__ __attribute__((aligned(16))) T sW[2 * SLF * 512];
  __shared__ __attribute__((aligned(16))) float sGam[T2_D];
  __shared__ float sAcc[2 * T2_D];
  __shared__ int sReady[T2_NFLAG], sDone[T2_NFLAG], sFin;

  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = a.K, nslab = K / D;
  const int ntiles = (a.M + 15) / 16, base = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  const int tile0 = (int)blockIdx.x * base + min((int)blockIdx.x, rem), ntile_wg = base + ((int)blockIdx.x < rem ? 1 : 0);
  const bool active = wave < ntile_wg;
  const int row = 16 * (tile0 + wave) + c;
  const int rowc = min(row, a.M - 1);
  const float valid = row < a.M ? 1.0f : 0.0f;

  if (threadIdx.x < T2_NFLAG) { sReady[threadIdx.x] = 0; sDone[threadIdx.x] = 0; }
  if (threadIdx.x == 0) sFin = 0;
  for (int i = threadIdx.x; i < 2 * D; i += T2_THREADS) sAcc[i] = 0.f;
  for (int i = threadIdx.x; i < D; i += T2_THREADS) sGam[i] = a.gamma[i];

  if (wave >= T2_WAVES) {      // loaders: half a slab each (36 pieces), slab s -> buffer s & 1
    __builtin_amdgcn_s_setprio(T2_LOADER_PRIO);
    const int half = wave == T2_WAVES ? 0 : SLF / 2;
    auto dma_half = [&](int sl) {
#pragma unroll
      for (int f = 0; f < SLF / 2; ++f) {
        const T* src = reinterpret_cast<const T*>(a.wt) + ((size_t)sl * SLF + half + f) * 512 + lane * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sW + ((sl & 1) * SLF + half + f) * 512), 16, 0, 0);
      }
    };
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("s_barrier" ::: "memory");
    dma_half(0);
    if (nslab > 1) dma_half(1);
    for (int sl = 0; sl < nslab; ++sl) {
      if (sl + 1 < nslab) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");   // only slab sl + 1's pieces may be in flight
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      if (lane == 0) atomicAdd(&sReady[sl], 1);
      if (sl + 2 < nslab) {
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sDone[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < ntile_wg)
          __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
        dma_half(sl + 2);
      }
    }
    return;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  asm volatile("s_barrier" ::: "memory");
  if (!active) return;

  const T* const dr = reinterpret_cast<const T*>(a.dy) + (size_t)rowc * K + 8 * g;
  Frag<T> bfa[KS], bfb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bfa[ks] = ld_frag(dr + 32 * ks);
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto chunk = [&](int sl, const Frag<T> (&cur)[KS], Frag<T> (&nxt)[KS]) {
    if (sl + 1 < nslab) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) nxt[ks] = ld_frag(dr + (sl + 1) * D + 32 * ks);
    }
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&sReady[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 2)
      __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    {   // a 192-deep slab = three 64-deep k chunks of 24 fragments ((tile * 2 + k step) inside a chunk)
      const T* wb = sW + (sl & 1) * SLF * 512 + lane * 8;
#pragma unroll
      for (int kcl = 0; kcl < 3; ++kcl) t2_gemm<NT, 2, 8>(wb + kcl * T2_HALF * 512, cur + 2 * kcl, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&sDone[sl], 1);
  };
  for (int sl = 0; sl < nslab; sl += 2) {
    chunk(sl, bfa, bfb);
    if (sl + 1 < nslab) chunk(sl + 1, bfb, bfa);
  }
  t2_ln_backward(acc, reinterpret_cast<const T*>(a.x) + (size_t)rowc * D + 4 * g,
                 reinterpret_cast<const T*>(a.dres) + (size_t)rowc * D + 4 * g, a.mean[rowc], a.rstd[rowc], sGam, sAcc, c, g,
                 valid, t2_row(a.dx, 16 * tile0, rowc, D));
  t2_flush_colsums(sAcc, &sFin, ntile_wg, lane, a.dgamma, a.dbeta);
}

// Fragment-major packed copy of a weight matrix W [R, C] (R % 16 == 0, C % kchunk == 0, kchunk % 32 == 0) for the kernel
// above: block (kc, nt, ks) = 64 lanes x 8 elements at index ((kc * R/16 + nt) * KSC + ks), kc = kchunk-wide k chunk,
// KSC = kchunk / 32, ks = 32-deep step inside it; lane l = 16g + cc, element e  <-  W[16nt + cc][kchunk kc + 32ks + k(g, e)]
//   natural: k = 8g + e            phi: k = e < 4 ? 4g + e : 16 + 4g + e - 4   (the acc_to_frag order)
template <typename T>
__global__ void pack_frags_kernel(const float* __restrict__ w, T* __restrict__ dst, int R, int Cc, int kchunk, int phi) {
  const long long total = (long long)R * Cc;
  const int NTr = R / 16, KSC = kchunk / 32;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63);
    long long blk = idx >> 9;
    const int ks = (int)(blk % KSC); blk /= KSC;
    const int nt = (int)(blk % NTr);
    const int kc = (int)(blk / NTr);
    const int cc = l & 15, g = l >> 4;
    const int k = phi ? (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4) : 8 * g + e;
    dst[idx] = from_f32<T>(w[(size_t)(16 * nt + cc) * Cc + (size_t)kchunk * kc + 32 * ks + k]);
  }
}

}  // namespace vitpe

using namespace vitpe;

extern "C" int vitpe_pack_weight_frags(int dtype, const float* w, void* packed, int R, int C, int kchunk, int phi,
                                       hipStream_t stream) {
  VITPE_REQUIRE(w && packed && R > 0 && C > 0 && kchunk > 0 && R % 16 == 0 && kchunk % 32 == 0 && C % kchunk == 0 &&
                (dtype == 0 || dtype == 1));
  const long long total = (long long)R * C;
  const unsigned blocks = (unsigned)min((total + 255) / 256, (long long)2048);
  if (dtype == 1) hipLaunchKernelGGL(pack_frags_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, w, (bf16*)packed, R, C, kchunk, phi);
  else hipLaunchKernelGGL(pack_frags_kernel<float>, dim3(blocks), dim3(256), 0, stream, w, (float*)packed, R, C, kchunk, phi);
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_block_tail2_supported(int dtype, int D, int HID) {
  return dtype == 1 && D == T2_D && HID >= 2 * T2_CH && HID % T2_CH == 0 && HID <= T2_MAXHID;
}

// The three weights are vitpe_pack_weight_frags copies: Wp (kchunk 192, natural), W1 (kchunk 192, phi), W2 (kchunk 32,
// phi); gp_out = gelu'(u), not u.  gp_out and h_out: both or neither.
static int tail2_launch(int dtype, const void* attn_out, const void* x_in, const void* Wp_packed, const float* bp,
                        const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                        void* xn_out, const void* W1_packed, const float* b1, const void* W2_packed,
                        const float* b2, void* gp_out, void* h_out, void* out, float* mean_out, float* rstd_out,
                        float eps2, float eps_next, int M, int D, int HID, unsigned long long* census, int exp, hipStream_t stream) {
  VITPE_REQUIRE(attn_out && x_in && Wp_packed && bp && gamma && beta && x_mid && mean2 && rstd2 && W1_packed && b1 &&
                W2_packed && b2 && out && M >= 0);
  VITPE_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr) && (gp_out == nullptr) == (h_out == nullptr));
  if (!vitpe_block_tail2_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  Tail2Args a{};
  a.a = attn_out; a.xin = x_in; a.wp = Wp_packed; a.bp = bp; a.gamma = gamma; a.beta = beta; a.w1 = W1_packed; a.b1 = b1;
  a.w2 = W2_packed; a.b2 = b2; a.xmid = x_mid; a.mean2 = mean2; a.rstd2 = rstd2; a.xn_out = xn_out; a.gp_out = gp_out;
  a.h_out = h_out; a.out = out; a.mean_out = mean_out; a.rstd_out = rstd_out; a.M = M; a.HID = HID; a.eps2 = eps2;
  a.eps_next = eps_next;
  // 16-token tiles over workgroups of <= 9 waves: as many workgroups as CUs (x rounds), 8 tiles each where that fits
  const int ntiles = (M + 15) / 16;
  int grid;
  if (ntiles <= 256 * 8) grid = (ntiles + 7) / 8;
  else grid = 256 * ((ntiles + 256 * T2_WAVES - 1) / (256 * T2_WAVES));
  a.census = census;
  if (census != nullptr) {
    VITPE_REQUIRE(gp_out != nullptr);
    if (exp == 1) hipLaunchKernelGGL((block_tail2_fwd_kernel<true, true, 1>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
    else if (exp == 2) hipLaunchKernelGGL((block_tail2_fwd_kernel<true, true, 2>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
    else if (exp == 4) hipLaunchKernelGGL((block_tail2_fwd_kernel<true, true, 4>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
    else if (exp == 7) hipLaunchKernelGGL((block_tail2_fwd_kernel<true, true, 7>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((block_tail2_fwd_kernel<true, true>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
  } else if (gp_out != nullptr) {
    hipLaunchKernelGGL((block_tail2_fwd_kernel<true, false>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
  } else {
    hipLaunchKernelGGL((block_tail2_fwd_kernel<false, false>), dim3(grid), dim3(T2F_THREADS), 0, stream, a);
  }
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_block_tail2_fwd(int dtype, const void* attn_out, const void* x_in, const void* Wp_packed, const float* bp,
                                     const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                                     void* xn_out, const void* W1_packed, const float* b1, const void* W2_packed,
                                     const float* b2, void* gp_out, void* h_out, void* out, float* mean_out, float* rstd_out,
                                     float eps2, float eps_next, int M, int D, int HID, hipStream_t stream) {
  return tail2_launch(dtype, attn_out, x_in, Wp_packed, bp, gamma, beta, x_mid, mean2, rstd2, xn_out, W1_packed, b1, W2_packed,
                      b2, gp_out, h_out, out, mean_out, rstd_out, eps2, eps_next, M, D, HID, nullptr, 0, stream);
}

// debug (include/vitpe_debug.h): the training instantiation with s_memtime stamps, census[(workgroup * 9 + wave) * 16 + slot]
extern "C" int vitpe_debug_tail2_census(const void* attn_out, const void* x_in, const void* Wp_packed, const float* bp,
                                        const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                                        void* xn_out, const void* W1_packed, const float* b1, const void* W2_packed,
                                        const float* b2, void* gp_out, void* h_out, void* out, float* mean_out,
                                        float* rstd_out, int M, int HID, unsigned long long* census, int exp, hipStream_t stream) {
  VITPE_REQUIRE(census != nullptr);
  return tail2_launch(1, attn_out, x_in, Wp_packed, bp, gamma, beta, x_mid, mean2, rstd2, xn_out, W1_packed, b1, W2_packed, b2,
                      gp_out, h_out, out, mean_out, rstd_out, 1e-5f, 1e-5f, M, T2_D, HID, census, exp, stream);
}

// Backward of vitpe_block_tail2_fwd w.r.t. its inputs: du [M,HID] (fc1's weight gradient reads it), dx_mid [M,192], da [M,192]
// (the attention backward's input); dgamma / dbeta of norm2 accumulated.  Weights as packed copies of the TRANSPOSES:
// W2t_packed = pack(fc2.weight^T [HID,192], 192, 1), W1t_packed = pack(fc1.weight^T [192,HID], 32, 1),
// WpT_packed = pack(attn.proj.weight^T [192,192], 192, 1).
static int tail2_bwd_launch(const Tail2BwdArgs& a0, bool pre, hipStream_t stream) {
  Tail2BwdArgs a = a0;
  const int ntiles = (a.M + 15) / 16;
  int grid;
  if (ntiles <= 256 * 8) grid = (ntiles + 7) / 8;
  else grid = 256 * ((ntiles + 256 * T2_WAVES - 1) / (256 * T2_WAVES));
  if (pre) hipLaunchKernelGGL(block_tail2_bwd_kernel<true>, dim3(grid), dim3(T2_THREADS), 0, stream, a);
  else hipLaunchKernelGGL(block_tail2_bwd_kernel<false>, dim3(grid), dim3(T2_THREADS), 0, stream, a);
  VITPE_CHECK_LAUNCH();
}
extern "C" int vitpe_block_tail2_bwd(int dtype, const void* dy, const void* gp, const void* W2t_packed, const void* W1t_packed,
                                     const void* x_mid, const float* mean2, const float* rstd2, const float* gamma, void* du,
                                     void* dx_mid, float* dgamma, float* dbeta, const void* WpT_packed, void* da, int M, int D,
                                     int HID, hipStream_t stream) {
  VITPE_REQUIRE(dy && gp && W2t_packed && W1t_packed && x_mid && mean2 && rstd2 && gamma && du && dx_mid && dgamma && dbeta &&
                WpT_packed && da && M >= 0);
  if (!vitpe_block_tail2_supported(dtype, D, HID)) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  Tail2BwdArgs a{};
  a.dy = dy; a.gp = gp; a.xmid = x_mid; a.mean2 = mean2; a.rstd2 = rstd2; a.gamma = gamma; a.w2t = W2t_packed; a.w1t = W1t_packed;
  a.wpt = WpT_packed; a.du = du; a.dxmid = dx_mid; a.da = da; a.dgamma = dgamma; a.dbeta = dbeta; a.M = M; a.HID = HID;
  return tail2_bwd_launch(a, false, stream);
}
// The same with the block ABOVE's qkv data gradient + LayerNorm1 backward + residual (vitpe_linear_lnbwd2's function) run
// first in the same kernel: dy_out = dres1 + LayerNorm1'(d_qkv Wqkv) is WRITTEN (the weight gradients read it) and feeds the
// MLP backward from registers.  WqT_packed = pack(attn.qkv.weight^T [192,K1], kchunk 64, phi 0) of the upper block; x1 / mean1 /
// rstd1 / gamma1 its LayerNorm1 input rows, statistics and weight; dgamma1 / dbeta1 accumulated.  K1 % 64 == 0, K1 <= 640.
extern "C" int vitpe_block_tail2_bwd_pre(int dtype, const void* d_qkv, const void* WqT_packed, const void* x1, const float* mean1,
                                         const float* rstd1, const float* gamma1, const void* dres1, float* dgamma1,
                                         float* dbeta1, int K1, void* dy_out, const void* gp, const void* W2t_packed,
                                         const void* W1t_packed, const void* x_mid, const float* mean2, const float* rstd2,
                                         const float* gamma, void* du, void* dx_mid, float* dgamma, float* dbeta,
                                         const void* WpT_packed, void* da, int M, int D, int HID, hipStream_t stream) {
  VITPE_REQUIRE(d_qkv && WqT_packed && x1 && mean1 && rstd1 && gamma1 && dres1 && dgamma1 && dbeta1 && dy_out && gp &&
                W2t_packed && W1t_packed && x_mid && mean2 && rstd2 && gamma && du && dx_mid && dgamma && dbeta && WpT_packed &&
                da && M >= 0);
  if (!vitpe_block_tail2_supported(dtype, D, HID) || K1 <= 0 || K1 % 64 != 0 || K1 > 640) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  Tail2BwdArgs a{};
  a.dy = dy_out; a.gp = gp; a.xmid = x_mid; a.mean2 = mean2; a.rstd2 = rstd2; a.gamma = gamma; a.w2t = W2t_packed;
  a.w1t = W1t_packed; a.wpt = WpT_packed; a.du = du; a.dxmid = dx_mid; a.da = da; a.dgamma = dgamma; a.dbeta = dbeta; a.M = M;
  a.HID = HID; a.dqkv = d_qkv; a.wqt = WqT_packed; a.x1 = x1; a.mean1 = mean1; a.rstd1 = rstd1; a.gamma1 = gamma1; a.dres1 = dres1;
  a.dgamma1 = dgamma1; a.dbeta1 = dbeta1; a.K1 = K1;
  return tail2_bwd_launch(a, true, stream);
}

// dx = dres + LayerNorm'(dY W) with Wt_packed = pack(W^T [192,K], kchunk 64, natural) (W = the Linear's weight [K,192], e.g.
// attn.qkv.weight): vitpe_linear_lnbwd on the wave-per-tile mapping.  bf16, K % 192 == 0, K <= 192 * 24.
extern "C" int vitpe_linear_lnbwd2(int dtype, const void* dY, const void* Wt_packed, void* dx, const void* x, const float* mean,
                                   const float* rstd, const float* gamma, const void* dres, float* dgamma, float* dbeta, int M,
                                   int K, hipStream_t stream) {
  VITPE_REQUIRE(dY && Wt_packed && dx && x && mean && rstd && gamma && dres && dgamma && dbeta && M >= 0);
  if (dtype != 1 || K <= 0 || K % T2_D != 0 || K / T2_D >= T2_NFLAG) return (int)hipErrorNotSupported;
  if (M == 0) return 0;
  LnBwd2Args a{};
  a.dy = dY; a.wt = Wt_packed; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.dres = dres; a.dx = dx;
  a.dgamma = dgamma; a.dbeta = dbeta; a.M = M; a.K = K;
  const int ntiles = (M + 15) / 16;
  int grid;
  if (ntiles <= 256 * 8) grid = (ntiles + 7) / 8;
  else grid = 256 * ((ntiles + 256 * T2_WAVES - 1) / (256 * T2_WAVES));
  hipLaunchKernelGGL(ln_bwd2_kernel, dim3(grid), dim3(T2_THREADS), 0, stream, a);
  VITPE_CHECK_LAUNCH();
}
