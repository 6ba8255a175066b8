// Fused attention-with-positional-encoding forward on 32x32 matrix-core tiles ("wide" kernel, round 3): the
// benchmark geometry N = 65 (64 patch tokens + class token), d = 192, 6 heads of 32, bf16.
//
// Replaces, per layer, reference models/vit.py:47-88 exactly as csrc/attn.hip does (qkv Linear without bias -> head
// split -> RoPE rotate-half on the patch tokens, rope_utils.py:18-37 / vit.py:56-68 -> QK^T * hd^-0.5 -> + relative /
// polynomial bias, positional_encoding.py:82-95 / 127-171 -> softmax -> @V -> merged heads), with the preceding
// LayerNorm (vit.py:113,122) optionally applied while the tokens are staged.
//
// Why a second forward kernel.  The 16x16x32 kernel of attn.hip pads 65 tokens to 80 and spends 5.6 vector
// instructions per MFMA; its counters say the vector issue port, not the matrix pipe, is what it runs out of (PMC,
// 25 % of the wave-cycles issue, 47 % wait for an issue slot).  A 16x16x32 MFMA holds that port for 8
// of its 16 cycles, a 32x32x16 MFMA for 8 of its 32 (MI355X_MICROARCH.md, issue-cost row) at twice the flop, and
// 64 = 2 x 32: on 32-wide tiles the patch tokens need NO padding.  So:
//   * tokens 0..63 (class token + 63 patches, natural order) are two 32-token tiles; token 64 is the "odd" token;
//   * one wave per (image, head) as before, two images per workgroup, nothing of q/k/v in LDS;
//   * V is projected un-swapped (A = x rows, B = Wv): the accumulators ARE the V^T operand of O^T = V^T P^T;
//     K and Q swapped (A = W rows, B = x rows): the accumulators are the operands of S^T = K Q^T, RoPE in registers
//     (the rotate-half partner f + 16 is register rho + 8 of the same lane);
//   * weight fragments live in a 12-deep register ring: k-step s of the NEXT matrix is requested as soon as k-step s of
//     the current one has been issued for both token tiles (a fragment is used twice, and requested a whole matrix
//     ahead: ~12 k-steps of L2 latency cover);
//   * the logits of a 32-query tile against all 64 keys are 2 x 16 registers per lane with the query on the lane:
//     max and sum are in-lane plus one v_permlane32_swap; P is packed to bf16 in place and is the B operand of P.V;
//   * the odd token: its q/k/v rows (2 images x 576 features) are projected ONCE per workgroup on 16x16x32 tiles
//     (18 MFMAs per wave instead of a padded fifth tile = 36), exchanged through LDS behind one barrier; as a KEY it is
//     one extra k-step whose A operand carries k_odd in every row; as a QUERY its 65 logits are produced with the keys
//     on the lanes (A = q_odd in every row, B = K^T), so its softmax costs a 32-lane reduction instead of a whole tile.
// Weights come from the "wide" pack (vitpe_pack_qkv_weights_wide): section 0 the 32x32x16 fragments, section 1 the
// 16x16x32 fragments of the odd-token projection; the q rows are pre-multiplied by hd^-0.5 * log2(e) so that the
// logits come out of the matrix core in the exp2 domain and no instruction scales q.
#include "attn_common.h"
#include <stdlib.h>

// cache policy of the output stores (raw buffer stores): 0 plain, 2 nt, 16 sc1 = write-through
#ifndef VITPE_A32_STORE_AUX
#define VITPE_A32_STORE_AUX 16
#endif
#ifndef VITPE_A32_STORE_AUX_XN
#define VITPE_A32_STORE_AUX_XN 16
#endif

namespace vitpe {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct W32 {
  static constexpr int D = 192, HD = 32, H = 6, N = 65, P = 64, KS = 12, NTH = 768;
  static constexpr int LDX = 200;    // token rows: 400 B = 25 16-B slots (odd): the 16 lanes a ds_read_b128 is served in hit 16 slots
  static constexpr int CSLD = 20;    // cos / sin rows: 80 B = 5 slots (odd), same reason
  static constexpr int XIMG = N * LDX;
  static constexpr int WSCR = 256;   // bf16 elements of per-wave scratch: q_odd 0.., k_odd 32.., v_odd 64.., P row 128..
};

VITPE_DEV void mma32(const bf16x8& a, const bf16x8& b, f32x16& c) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// registers 8s .. 8s+7 of a 32x32 accumulator tile as the operand fragment of k-step s of the next product
// (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): position (lane half hh, element j) <->
// accumulator row 16 s + 4 hh + (j & 3) + 8 (j >> 2) on BOTH operands of that product
template <int S>
VITPE_DEV bf16x8 pack8(const f32x16& a) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16)a[8 * S + j];
  return f;
}

VITPE_DEV bf16x8 ldsfrag(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
// (fmaxf on matrix-core outputs costs a canonicalising v_max per operand; the logits are finite by construction)
VITPE_DEV float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// the same LDS address read again in a later phase must be a NEW read: left alone, the compiler keeps all 24 token
// fragments of the first projection alive for the other two (96 registers) and spills
VITPE_DEV const bf16* fresh(const bf16* p) { asm volatile("" : "+v"(p)); return p; }

// max / sum over the 32 lanes of a half-wave (both halves hold the same data): xor 1, 2 by quad_perm, 4 and 8 by
// row rotations of the 16-lane DPP rows, 16 by v_permlane16_swap
#define VITPE_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
VITPE_DEV float half_max(float v) {
  v = fmaxf(v, VITPE_DPP(v, 0xB1));    // quad_perm [1,0,3,2]
  v = fmaxf(v, VITPE_DPP(v, 0x4E));    // quad_perm [2,3,0,1]
  v = fmaxf(v, VITPE_DPP(v, 0x124));   // row_ror:4
  v = fmaxf(v, VITPE_DPP(v, 0x128));   // row_ror:8
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
VITPE_DEV float half_sum(float v) {
  v += VITPE_DPP(v, 0xB1);
  v += VITPE_DPP(v, 0x4E);
  v += VITPE_DPP(v, 0x124);
  v += VITPE_DPP(v, 0x128);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
VITPE_DEV float swap32_max(float v) {
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
VITPE_DEV float swap32_sum(float v) {
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// KM: positional-encoding class (attn_common.h); LNF: LayerNorm fused into the staging; MIXED: per-head cos / sin
// tables (rope-mixed); CENSUS: debug stamps
//
// Projection = a k-loop over six 32-deep chunks, synchronised PER HEAD.  The two waves of a head (its two images) share
// a private double-buffered ring of that head's Wq / Wk / Wv chunk (6 1-KB fragments per chunk, filled by LDS-DMA: three
// pieces per wave, a chunk ahead; every weight byte leaves L2 once per workgroup -- the register-ring version fetched
// 648 KB per CU from L2, each wave its own copy, and ran at the L2 -> CU rate).  Per 16-deep k-step a wave reads two token
// fragments and its head's three weight fragments and issues six MFMAs into SIX live accumulators (v, k, q of both token
// tiles: 96 registers): every operand byte is read from LDS exactly once.  The odd token's 16x16x32 operands are
// gathered from the same LDS image (a 16x16x32 fragment is a different lane -> address map of two consecutive 32x32x16
// fragments): the pair projects the odd tokens of BOTH images for its head, wave i the 16-feature tile i of q, k, v.
// Synchronisation after the staging barrier is a two-wave handshake per chunk (an LDS counter per head and chunk: "my
// reads of chunk c are done and my pieces of chunk c + 1 have landed"): no workgroup barrier, so the three waves of a
// SIMD drift apart and one's LDS waits and softmax run under another's MFMAs (with s_barrier per chunk the oldest wave
// of a SIMD spent 6 K of its 14 K k-loop cycles parked: census in DESIGN.md).  The LDS-DMA is issued from inline asm:
// through the builtin, hipcc waits vmcnt(0) right behind the issue (it cannot tell the ring slots apart).
template <int KM, bool LNF, bool MIXED, bool CENSUS = false, int EXP = 0>   // EXP: timing experiments of the census build (WRONG results)
__global__ __launch_bounds__(768, 3) void attn32_fwd_kernel(AttnArgs a) {
  using C = AttnCfg<bf16, 32, 192, 5, 1, 65>;   // (bias-table helpers of attn_common.h: TABLD = 160, PBLD = 32)
  constexpr int D = W32::D, N = W32::N, LDX = W32::LDX, H = W32::H;
  constexpr int CSLD = MIXED ? 16 : W32::CSLD;   // (mixed: unpadded rows, or the six tables would not fit the token region)
  constexpr int PBUF = 6 * 512, HBUF = 2 * PBUF;   // a head's ring: 2 buffers x (matrix, k-step in chunk) x 1 KB
  // rope-mixed: the per-head tables (62 KB) do not fit beside the weight ring; they are staged into the token region
  // once the projections are done with it
  constexpr int CSTAB = (KM != KM_ROPE) ? 4 : (MIXED ? 4 : N * CSLD);
  static_assert(!MIXED || H * N * CSLD * 4 * 2 <= 2 * W32::XIMG * 2, "mixed tables must fit the token region");
  __shared__ __attribute__((aligned(16))) bf16 xs_all[2 * W32::XIMG];
  __shared__ __attribute__((aligned(16))) bf16 wbuf[H * HBUF];
  __shared__ int pflag[H * 8];   // [head][handshake]: arrivals (2 = both waves)
  __shared__ __attribute__((aligned(16))) float s_cos_ax[CSTAB];   // [token][CSLD]; token 0 = identity
  __shared__ __attribute__((aligned(16))) float s_sin_ax[CSTAB];
  __shared__ __attribute__((aligned(16))) float odd_raw[2 * 3 * D];   // [image][head][q|k|v][32] of token 64
  __shared__ __attribute__((aligned(16))) bf16 wscr[12 * W32::WSCR];
  __shared__ __attribute__((aligned(16))) float s_tab[KM == KM_RELATIVE ? C::H * C::TABLD : 4];
  __shared__ __attribute__((aligned(16))) float s_coef[KM == KM_POLY ? C::PESZ : 4];
  float* const s_cos = MIXED ? reinterpret_cast<float*>(xs_all) : s_cos_ax;
  float* const s_sin = MIXED ? reinterpret_cast<float*>(xs_all) + H * N * CSLD : s_sin_ax;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wave & 1, h = wave >> 1;
  const int b_raw = blockIdx.x * 2 + img;
  const bool live = b_raw < a.B;                 // odd batch: the last workgroup's second image is a copy that stores nothing
  const int b = live ? b_raw : a.B - 1;
  const int r = lane & 31, hh = lane >> 5;
  auto stamp = [&](int slot) {
    if (CENSUS) {
      __builtin_amdgcn_sched_barrier(0);
      if (lane == 0) a.census[((size_t)blockIdx.x * 16 + wave) * 16 + slot] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
  if (CENSUS && lane == 0) a.census[((size_t)blockIdx.x * 16 + wave) * 16 + 9] = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_bar = 0, t_dma = 0;
  const bf16* const Wbase = reinterpret_cast<const bf16*>(a.wqkv);
  // chunk c of this head -> ring buffer buf: piece p = matrix * 2 + k-step-in-chunk; this wave issues p = 3 img .. + 2
  const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(wbuf + h * HBUF);
  const unsigned lane16 = lane * 16;
  auto dma_chunk = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int p = 3 * img + i, mat = p >> 1, sp = p & 1;
      const bf16* src = Wbase + (size_t)((h * 3 + mat) * 12 + 2 * c + sp) * 512;        // wave-uniform
      const unsigned dst = ring_lds + (unsigned)(buf * PBUF + p * 512) * 2;              // LDS byte address, wave-uniform
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(src) : "memory");
    }
  };
  if (tid < H * 8) pflag[tid] = 0;
  dma_chunk(0, 0);
  dma_chunk(1, 1);

  // ---- stage both images' tokens (LayerNorm on the way in), the PE tables
  {
    // the workgroup's two images are 130 consecutive rows: one wave-uniform base, 32-bit per-thread offsets (the
    // per-image base pointers of the first version cost 64-bit address arithmetic per load: a third of this phase's VALU)
    const int simg = tid / 384, st = tid % 384, cc = st % 24, r0 = st / 24;
    const bool slive = blockIdx.x * 2 + simg < a.B;
    const int srow = slive ? simg * N : 0;                       // odd batch: the idle second slot re-reads image 0
    const bf16* const xbase = reinterpret_cast<const bf16*>(a.xn) + (size_t)blockIdx.x * 2 * N * D;
    const float* const mbase = LNF ? a.ln_mean + (size_t)blockIdx.x * 2 * N : nullptr;
    const float* const rbase = LNF ? a.ln_rstd + (size_t)blockIdx.x * 2 * N : nullptr;
    Chunk16 v[5];
    float mu[5], rs[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const unsigned rowg = (unsigned)(srow + min(r0 + 16 * it, N - 1));
      v[it] = *reinterpret_cast<const Chunk16*>(xbase + rowg * (unsigned)D + (unsigned)(cc * 8));
      mu[it] = LNF ? mbase[rowg] : 0.f;
      rs[it] = LNF ? rbase[rowg] : 0.f;
    }
    if (LNF) {
      float gq[8], bq[8];
      {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_gamma + cc * 8), g1 = *reinterpret_cast<const f32x4*>(a.ln_gamma + cc * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_beta + cc * 8), b1 = *reinterpret_cast<const f32x4*>(a.ln_beta + cc * 8 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { gq[t] = g0[t]; gq[4 + t] = g1[t]; bq[t] = b0[t]; bq[4 + t] = b1[t]; }
      }
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int row = r0 + 16 * it;
        if (it < 4 || row < N) {
          float f[8];
          chunk_to_f32<bf16>(v[it], f);
          const float sc = rs[it], sh = -mu[it] * rs[it];
#pragma unroll
          for (int t = 0; t < 8; ++t) f[t] = fmaf(fmaf(f[t], sc, sh), gq[t], bq[t]);
          v[it] = f32_to_chunk<bf16>(f);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int row = r0 + 16 * it;
      if (it < 4 || row < N) *reinterpret_cast<Chunk16*>(xs_all + simg * W32::XIMG + row * LDX + cc * 8) = v[it];
    }
    if (KM == KM_ROPE && !MIXED) {   // cos / sin -> LDS indexed by TOKEN; the class token's row is the identity rotation (vit.py:56-68)
      for (int q = tid; q < N * 4; q += W32::NTH) {
        const int f4 = q & 3, row = q >> 2;
        f32x4 cv = {1.f, 1.f, 1.f, 1.f}, sv = {0.f, 0.f, 0.f, 0.f};
        if (row >= 1) {
          cv = *reinterpret_cast<const f32x4*>(a.cos + (size_t)(row - 1) * 16 + 4 * f4);
          sv = *reinterpret_cast<const f32x4*>(a.sin + (size_t)(row - 1) * 16 + 4 * f4);
        }
        *reinterpret_cast<f32x4*>(&s_cos_ax[row * CSLD + 4 * f4]) = cv;
        *reinterpret_cast<f32x4*>(&s_sin_ax[row * CSLD + 4 * f4]) = sv;
      }
    }
    if (KM == KM_RELATIVE) {
      for (int q = tid; q < C::H * C::TABLD; q += W32::NTH) {
        const int hq = q / C::TABLD, i = q % C::TABLD;
        s_tab[q] = (i < 2 * N - 1) ? a.table[hq * (2 * N - 1) + i] * LOG2E : 0.f;
      }
    }
    if (KM == KM_POLY) stage_poly<C>(a, 0, s_coef, N, tid, W32::NTH);
  }
  stamp(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the six LDS-DMA pieces of chunks 0 and 1 (hipcc does not count them)
  __syncthreads();
  stamp(2);

  // ---- the k-loop
  const bf16* const xr0 = xs_all + img * W32::XIMG + r * LDX + 8 * hh;   // token tile 0, k-step s: + 16 s
  const bf16* const xr1 = xr0 + 32 * LDX;
  const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 aV0 = z16, aV1 = z16, aK0 = z16, aK1 = z16, aQ0 = z16, aQ1 = z16;
  // odd token (row 64 of both images; image = column & 1): this wave's feature tiles are nt = img of q, k, v of its head
  // (flat feature 16 ft = head * 96 + {q,k,v} * 32 + 16 nt); lane (c, g) of the 16x16x32 A operand <- lane
  // (16 nt + c) + 32 (g & 1) of the 32x32x16 fragment of k-step (g >> 1) of the chunk
  const int oc = lane & 15, og = lane >> 4;
  const bf16* const xodd = xs_all + (oc & 1) * W32::XIMG + 64 * LDX + 8 * og;
  int ooff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    ooff[t] = h * HBUF + (t * 2 + (og >> 1)) * 512 + ((16 * img + oc) + 32 * (og & 1)) * 8;
  }
  f32x4 od[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // Rolling prefetch: every LDS read is issued two MFMAs (>= 64 matrix-pipe cycles of this wave, ~200 with its two SIMD
  // partners) ahead of its first use, never more than ~8 fragments in flight (the six accumulators leave no room for a
  // whole k-step of double buffering).  Order per k-step: v (needs x, Wv), k (Wk), q (Wq); the next k-step's x / Wv / Wk
  // are requested under them.  The next chunk's token fragments (static data) are requested BEFORE the barrier, its
  // weight fragments right after it.  Two chunks per loop iteration: every LDS address is base + immediate.
  const bf16* const wp = wbuf + h * HBUF + lane * 8;            // buffer 0, this head's q fragment of k-step 0; k: + 2 * 512, v: + 4 * 512
  const bf16* xp = xr0;                                          // k-step 0 of the current chunk pair
  const bf16* xop = xodd;
  // two-wave handshake k of this head: own LDS reads done + own LDS-DMA pieces landed, then wait for the partner
  auto pair_sync = [&](int k) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = 0;
    if (CENSUS) t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (EXP & 4) {
      __builtin_amdgcn_s_barrier();
    } else if (!(EXP & 2)) {
      typedef __attribute__((address_space(3))) int lds_int;
      lds_int* const ctr = (lds_int*)(pflag + h * 8 + k);
      int seen = 0;
      if (lane == 0) seen = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // 1: the partner is already here
      seen = __builtin_amdgcn_readfirstlane(seen);
      for (int spin = 0; seen < 1 && spin < (1 << 18); ++spin) {        // (bounded: a lost partner must not hang the chip)
        __builtin_amdgcn_s_sleep(1);
        seen = __builtin_amdgcn_readfirstlane(*(volatile lds_int*)ctr) - 1;
      }
    }
    asm volatile("" ::: "memory");
    if (CENSUS) t_bar += __builtin_amdgcn_s_memtime() - t0;
    __builtin_amdgcn_sched_barrier(0);
  };
  // One k-step of double buffering: the five fragments of k-step j + 1 are requested before the six MFMAs of k-step j
  // (a wave alone on the matrix pipe then covers ~190 cycles of LDS latency; two MFMAs of lead, the first version,
  // left the oldest wave of a SIMD at a third of the pipe rate).  The handshake of chunk c sits between its two k-steps:
  // all reads of chunk c are issued (and waited for) by then, so the pieces of chunk c + 2 may overwrite its buffer one
  // whole chunk before they are needed, and chunk c + 1 -- requested at the previous handshake -- has landed.
#define VITPE_A32_FENCE __builtin_amdgcn_sched_barrier(0);
#define VITPE_A32_MMA6(XA, XB, WV, WK, WQ)                                                                          \
  mma32(XA, WV, aV0); mma32(XB, WV, aV1); mma32(WK, XA, aK0); mma32(WK, XB, aK1); mma32(WQ, XA, aQ0); mma32(WQ, XB, aQ1);
#define VITPE_A32_CHUNK(C, BUF, XO)                                                                                 \
  {                                                                                                                 \
    const bf16* const wb = wp + (BUF) * PBUF;                                                                       \
    VITPE_A32_FENCE                                                                                                 \
    const bf16x8 xa1 = ldsfrag(xp + (XO) + 16), xb1 = ldsfrag(xp + (XO) + 16 + 32 * LDX);                           \
    const bf16x8 wv1 = ldsfrag(wb + 5 * 512), wk1 = ldsfrag(wb + 3 * 512), wq1 = ldsfrag(wb + 1 * 512);             \
    VITPE_A32_FENCE                                                                                                 \
    VITPE_A32_MMA6(xa, xb, wv, wk, wq)                                                                              \
    VITPE_A32_FENCE                                                                                                 \
    const bf16x8 xo = ldsfrag(xop + (XO));                                                                          \
    const bf16x8 wo0 = ldsfrag(wbuf + (BUF) * PBUF + ooff[0]), wo1 = ldsfrag(wbuf + (BUF) * PBUF + ooff[1]),        \
                 wo2 = ldsfrag(wbuf + (BUF) * PBUF + ooff[2]);                                                      \
    pair_sync(C);                                                                                                   \
    if ((C) + 2 < 6) dma_chunk((C) + 2, (BUF));                                                                     \
    if (!(EXP & 1)) {                                                                                               \
      od[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo0, xo, od[0], 0, 0, 0);                                     \
      od[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo1, xo, od[1], 0, 0, 0);                                     \
      od[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo2, xo, od[2], 0, 0, 0);                                     \
    }                                                                                                               \
    VITPE_A32_FENCE                                                                                                 \
    xa = ldsfrag(xp + (XO) + 32); xb = ldsfrag(xp + (XO) + 32 + 32 * LDX);   /* (past the last chunk: in-bounds, unused) */ \
    wv = ldsfrag(wp + (1 - (BUF)) * PBUF + 4 * 512); wk = ldsfrag(wp + (1 - (BUF)) * PBUF + 2 * 512);               \
    wq = ldsfrag(wp + (1 - (BUF)) * PBUF + 0 * 512);                                                                \
    VITPE_A32_FENCE                                                                                                 \
    VITPE_A32_MMA6(xa1, xb1, wv1, wk1, wq1)                                                                         \
    VITPE_A32_FENCE                                                                                                 \
  }
  bf16x8 xa = ldsfrag(xp), xb = ldsfrag(xp + 32 * LDX);
  bf16x8 wv = ldsfrag(wp + 4 * 512), wk = ldsfrag(wp + 2 * 512), wq = ldsfrag(wp + 0 * 512);
  // matrix-pipe-bound phase at raised priority: once the oldest wave of a SIMD is in its (VALU-bound) softmax, the younger
  // waves' MFMAs must still win the issue port -- the softmax fills the 24 free issue cycles of every 32-cycle MFMA
  if (!(EXP & 8)) __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
  for (int cp = 0; cp < 3; ++cp) {
    VITPE_A32_CHUNK(2 * cp, 0, 0)
    VITPE_A32_CHUNK(2 * cp + 1, 1, 32)
    xp += 64;
    xop += 64;
  }
#undef VITPE_A32_CHUNK
#undef VITPE_A32_MMA6
#undef VITPE_A32_FENCE
  if (oc < 2) {
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(&odd_raw[oc * 3 * D + 16 * ((h * 3 + t) * 2 + img) + 4 * og]) = od[t];
  }
  pair_sync(6);   // the odd token's rows are published to the partner
  if (!(EXP & 8)) __builtin_amdgcn_s_setprio(0);
  auto write_xn_out = [&]() {
    // LayerNorm(x) for the backward pass, from the LDS image AFTER the k-loop: stored from the staging phase the 12.8 MB
    // sat in front of every LDS-DMA piece in the in-order vmcnt queue, and the first handshakes waited for HBM writes
    if (LNF && a.xn_out != nullptr) {
      const int simg = tid / 384, st = tid % 384, cc = st % 24, r0 = st / 24;
      if (blockIdx.x * 2 + simg < a.B) {
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<bf16*>(a.xn_out) + (size_t)blockIdx.x * 2 * N * D, 0, 2 * N * D * 2, 0x00020000);
#pragma unroll
        for (int it = 0; it < 5; ++it) {
          const int row = r0 + 16 * it;
          if (it < 4 || row < N) {
            const Chunk16 c16 = *reinterpret_cast<const Chunk16*>(xs_all + simg * W32::XIMG + row * LDX + cc * 8);
            __builtin_amdgcn_raw_buffer_store_b128(c16, xrsrc, (int)(((simg * N + row) * D + cc * 8) * 2), 0, VITPE_A32_STORE_AUX_XN);
          }
        }
      }
    }
  };
  if (!MIXED) write_xn_out();
  stamp(3);

  // ---- v: acc[rho] = v[token 32 t + perm(rho, hh)][feature r] -> V^T operand fragments
  bf16x8 vf[4];
  vf[0] = pack8<0>(aV0); vf[1] = pack8<1>(aV0); vf[2] = pack8<0>(aV1); vf[3] = pack8<1>(aV1);
  if (MIXED) {   // per-head tables -> the token region (free once EVERY wave is through its k-loop), indexed by token
    write_xn_out();
    __syncthreads();
    for (int q = tid; q < H * N * 4; q += W32::NTH) {
      const int f4 = q & 3, row = (q >> 2) % N, t = (q >> 2) / N;
      f32x4 cv = {1.f, 1.f, 1.f, 1.f}, sv = {0.f, 0.f, 0.f, 0.f};
      if (row >= 1) {
        const size_t src = ((size_t)t * (N - 1) + row - 1) * 16 + 4 * f4;
        cv = *reinterpret_cast<const f32x4*>(a.cos + src);
        sv = *reinterpret_cast<const f32x4*>(a.sin + src);
      }
      *reinterpret_cast<f32x4*>(&s_cos[(t * N + row) * CSLD + 4 * f4]) = cv;
      *reinterpret_cast<f32x4*>(&s_sin[(t * N + row) * CSLD + 4 * f4]) = sv;
    }
    __syncthreads();
  }
  const int cs_tab = MIXED ? h * N * CSLD : 0;
  bf16* const ws = wscr + wave * W32::WSCR;
  {  // rotate the odd token's q and k (token 64 = grid position 63), park q | k | v as bf16 in this wave's scratch
    const float* raw = odd_raw + img * 3 * D + h * 96;
    const int m = lane >> 5, f = lane & 31, fl = f & 15;        // lanes 0-31: q, 32-63: k
    float val;
    if (KM == KM_ROPE) {
      const float x1 = raw[m * 32 + fl], x2 = raw[m * 32 + fl + 16];
      const float cv = s_cos[cs_tab + 64 * CSLD + fl], sv = s_sin[cs_tab + 64 * CSLD + fl];
      val = (f < 16) ? x1 * cv - x2 * sv : x1 * sv + x2 * cv;
    } else {
      val = raw[m * 32 + f];
    }
    ws[m * 32 + f] = (bf16)val;
    if (lane < 32) ws[64 + lane] = (bf16)raw[64 + lane];
  }

  // ---- k and q: acc[rho] = k[token 32 t + r][feature perm(rho, hh)]; rotate; -> fragments
  auto rope = [&](f32x16& acc, int tile) {
    if (KM == KM_ROPE) {
      const int tok = 32 * tile + r;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(&s_cos[cs_tab + tok * CSLD + 4 * hh]);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(&s_cos[cs_tab + tok * CSLD + 8 + 4 * hh]);
      const f32x4 n0 = *reinterpret_cast<const f32x4*>(&s_sin[cs_tab + tok * CSLD + 4 * hh]);
      const f32x4 n1 = *reinterpret_cast<const f32x4*>(&s_sin[cs_tab + tok * CSLD + 8 + 4 * hh]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {   // feature f = (q & 3) + 8 (q >> 2) + 4 hh < 16 pairs with f + 16 = register q + 8
        const float cv = (q < 4) ? c0[q & 3] : c1[q & 3], sv = (q < 4) ? n0[q & 3] : n1[q & 3];
        const float x1 = acc[q], x2 = acc[q + 8];
        acc[q] = x1 * cv - x2 * sv;
        acc[q + 8] = x1 * sv + x2 * cv;
      }
    }
  };
  bf16x8 kf[2][2], qf[2][2];
  rope(aK0, 0);
  rope(aK1, 1);
  kf[0][0] = pack8<0>(aK0); kf[0][1] = pack8<1>(aK0); kf[1][0] = pack8<0>(aK1); kf[1][1] = pack8<1>(aK1);
  rope(aQ0, 0);
  rope(aQ1, 1);
  qf[0][0] = pack8<0>(aQ0); qf[0][1] = pack8<1>(aQ0); qf[1][0] = pack8<0>(aQ1); qf[1][1] = pack8<1>(aQ1);
  stamp(4);

  // ---- the odd token's operand fragments from the wave scratch: position (hh, j) <-> feature 16 s + 4 hh + (j & 3) + 8 (j >> 2)
  auto odd_rows = [&](int base, bf16x8 (&f)[2]) {   // every row (A operand) / column (B operand) = the odd token's vector
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ws + base + 16 * s + 4 * hh);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(ws + base + 16 * s + 8 + 4 * hh);
      f[s] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  bf16x8 kfo[2];
  odd_rows(32, kfo);
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
  bf16x8 vfo;   // A operand [feature r][k position]: v_odd at position (hh 0, j 0), zero elsewhere
  {
    const bf16 vv = ws[64 + r];
#pragma unroll
    for (int j = 0; j < 8; ++j) vfo[j] = (bf16)0.f;
    vfo[0] = (hh == 0) ? vv : (bf16)0.f;
  }

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
  // the workgroup's two images' output rows as one buffer (wave-uniform descriptor, 32-bit byte offsets)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<bf16*>(a.out) + (size_t)blockIdx.x * 2 * N * D, 0, 2 * N * D * 2, 0x00020000);
  const unsigned obase = (unsigned)((img * N * D + h * 32) * 2);
  // O^T accumulators -> scaled bf16 rows: lane (query r, hh) holds features 8 m + 4 hh .. + 3 in registers 4 m .. 4 m + 3;
  // one v_permlane32_swap per dword pairs the halves' 8-B pieces into 16 contiguous bytes (features 16 mp + 8 hh .. + 7)
  auto store_rows = [&](const f32x16& o, float inv, int token, bool pred) {
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
      uint32_t d0[2], d1[2];
#pragma unroll
      for (int w2 = 0; w2 < 2; ++w2) {
        bf16x2 pa, pb;
        pa[0] = (bf16)(o[8 * mp + 2 * w2] * inv); pa[1] = (bf16)(o[8 * mp + 2 * w2 + 1] * inv);           // piece m = 2 mp
        pb[0] = (bf16)(o[8 * mp + 4 + 2 * w2] * inv); pb[1] = (bf16)(o[8 * mp + 4 + 2 * w2 + 1] * inv);   // piece m = 2 mp + 1
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, pa), __builtin_bit_cast(uint32_t, pb), false, false);
        d0[w2] = sw[0]; d1[w2] = sw[1];
      }
      if (pred) __builtin_amdgcn_raw_buffer_store_b128((u32x4s){d0[0], d0[1], d1[0], d1[1]}, orsrc,
                                                        (int)(obase + (unsigned)((token * D + 16 * mp + 8 * hh) * 2)), 0, VITPE_A32_STORE_AUX);
    }
  };

  // ---- per 32-query tile: S^T = K Q^T (+ bias), softmax in the exp2 domain, O^T = V^T P^T, store
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    __builtin_amdgcn_sched_barrier(0);
    const int i = 32 * qt + r;
    f32x16 so = z16, s0 = z16, s1 = z16;
    mma32(kfo[0], qf[qt][0], so);
    mma32(kf[0][0], qf[qt][0], s0);
    mma32(kf[1][0], qf[qt][0], s1);
    mma32(kfo[1], qf[qt][1], so);
    mma32(kf[0][1], qf[qt][1], s0);
    mma32(kf[1][1], qf[qt][1], s1);
    float sodd = so[0];                              // every row of that tile is the odd key
    if (KM == KM_RELATIVE) {
      sodd += pe_bias2<C, KM>(a, s_tab, s_coef, h, i, 64, N);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = (q & 3) + 8 * (q >> 2) + 4 * hh;
        s0[q] += pe_bias2<C, KM>(a, s_tab, s_coef, h, i, j, N);
        s1[q] += pe_bias2<C, KM>(a, s_tab, s_coef, h, i, 32 + j, N);
      }
    }
    if (KM == KM_POLY) {
      // bias tabulated by L1 grid distance: the packed coordinates of this lane's query once, of its keys four at a time
      // (16-B LDS reads), one v_sad_u8 + one table read per logit; class-token row / column = 0 (positional_encoding.py:165-169)
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const unsigned* const xy = reinterpret_cast<const unsigned*>(s_coef + C::H * C::PBLD);
      const float* const ptab = s_coef + (a.coeff_per_head ? h : 0) * C::PBLD;
      const unsigned xyi = xy[i];
      const bool qcls = (qt == 0) && (r == 0);
      sodd += qcls ? 0.f : ptab[__builtin_amdgcn_sad_u8(xyi, xy[64], 0u)];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xy + 8 * q4 + 4 * hh), xb = *reinterpret_cast<const u32x4*>(xy + 32 + 8 * q4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float b0 = ptab[__builtin_amdgcn_sad_u8(xyi, xa[e], 0u)], b1 = ptab[__builtin_amdgcn_sad_u8(xyi, xb[e], 0u)];
          const bool kcls = (q4 == 0) && (e == 0) && (hh == 0);            // key 0 = the class token
          s0[4 * q4 + e] += (qcls || kcls) ? 0.f : b0;
          s1[4 * q4 + e] += qcls ? 0.f : b1;
        }
      }
    }
    float m = sodd;
#pragma unroll
    for (int q = 0; q < 16; ++q) m = max3(m, s0[q], s1[q]);
    m = swap32_max(m);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      s0[q] = __builtin_amdgcn_exp2f(s0[q] - m);
      s1[q] = __builtin_amdgcn_exp2f(s1[q] - m);
    }
    const float podd = __builtin_amdgcn_exp2f(sodd - m);
    bf16x8 pfo;
#pragma unroll
    for (int j = 0; j < 8; ++j) pfo[j] = (bf16)0.f;
    pfo[0] = (hh == 0) ? (bf16)podd : (bf16)0.f;     // position (hh 0, j 0) only: the row sum below counts every position
    const bf16x8 pf0 = pack8<0>(s0), pf1 = pack8<1>(s0), pf2 = pack8<0>(s1), pf3 = pack8<1>(s1);
    // Row sums on the matrix core: an all-ones A operand against the same P^T fragments gives sum_k P^T[k][query] in every
    // row -- the 32 in-lane adds and the cross-half exchange leave the vector issue port, which is what this phase runs
    // out of; the denominator is then the sum of the SAME bf16-rounded probabilities the numerator uses.
    f32x16 o = z16, lsum = z16;
    mma32(vf[0], pf0, o);
    mma32(ones, pf0, lsum);
    mma32(vf[1], pf1, o);
    mma32(ones, pf1, lsum);
    mma32(vf[2], pf2, o);
    mma32(ones, pf2, lsum);
    mma32(vf[3], pf3, o);
    mma32(ones, pf3, lsum);
    mma32(vfo, pfo, o);
    mma32(ones, pfo, lsum);
    const float l = lsum[0];
    store_rows(o, __builtin_amdgcn_rcpf(l), i, live);
  }
  stamp(5);

  // ---- the odd token as a query: logits with the KEY on the lane (A = q_odd in every row, B = K^T), softmax by a
  //      32-lane reduction, P through the wave scratch into the B-operand order, O^T row stored from lanes 0 and 32
  {
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 qfo[2];
    odd_rows(0, qfo);
    f32x16 t0 = z16, t1 = z16, t2 = z16;
    mma32(qfo[0], kf[0][0], t0);
    mma32(qfo[0], kf[1][0], t1);
    mma32(qfo[0], kfo[0], t2);
    mma32(qfo[1], kf[0][1], t0);
    mma32(qfo[1], kf[1][1], t1);
    mma32(qfo[1], kfo[1], t2);
    float a0 = t0[0], a1 = t1[0], a2 = t2[0];        // keys r, 32 + r, 64 (all rows equal)
    if (KM == KM_RELATIVE || KM == KM_POLY) {
      a0 += pe_bias2<C, KM>(a, s_tab, s_coef, h, 64, r, N);
      a1 += pe_bias2<C, KM>(a, s_tab, s_coef, h, 64, 32 + r, N);
      a2 += pe_bias2<C, KM>(a, s_tab, s_coef, h, 64, 64, N);
    }
    const float m = half_max(max3(a0, a1, a2));
    const float p0 = __builtin_amdgcn_exp2f(a0 - m), p1 = __builtin_amdgcn_exp2f(a1 - m), p2 = __builtin_amdgcn_exp2f(a2 - m);
    const float l = half_sum(p0 + p1) + p2;
    bf16* const pr = ws + 128;
    if (hh == 0) { pr[r] = (bf16)p0; pr[32 + r] = (bf16)p1; }
    __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's own LDS writes have landed
    bf16x8 pq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(pr + 16 * ks + 4 * hh);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(pr + 16 * ks + 8 + 4 * hh);
      pq[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    bf16x8 pfo;
#pragma unroll
    for (int j = 0; j < 8; ++j) pfo[j] = (bf16)0.f;
    pfo[0] = (bf16)p2;
    f32x16 o = z16;
    mma32(vf[0], pq[0], o);
    mma32(vf[1], pq[1], o);
    mma32(vf[2], pq[2], o);
    mma32(vf[3], pq[3], o);
    mma32(vfo, pfo, o);
    store_rows(o, __builtin_amdgcn_rcpf(l), 64, live && r == 0);
  }
  stamp(6);
  if (CENSUS && lane == 0) {
    a.census[((size_t)blockIdx.x * 16 + wave) * 16 + 8] = t_bar;
    a.census[((size_t)blockIdx.x * 16 + wave) * 16 + 11] = t_dma;
    a.census[((size_t)blockIdx.x * 16 + wave) * 16 + 10] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- the wide pack ------------------------------------------------------------------------------------------------
// 3 D D elements: block ((h * 3 + mat) * (D / 16) + s) = 64 lanes x 8: lane (r = l & 31, hh = l >> 5), element j =
//   W[mat D + 32 h + r][16 s + 8 hh + j]                       (32x32x16 operand fragments, 1 KB each)
// with the q rows (mat 0) multiplied by qscale = hd^-0.5 * log2(e)
VITPE_DEV void wide_src(long long idx, int D, int& row, int& col, int& mat) {
  const int e = (int)(idx & 7), l = (int)((idx >> 3) & 63);
  long long blk = idx >> 9;
  const int S = D / 16;
  const int s = (int)(blk % S); blk /= S;
  mat = (int)(blk % 3);
  const int h = (int)(blk / 3);
  row = mat * D + 32 * h + (l & 31);
  col = 16 * s + 8 * (l >> 5) + e;
}
__global__ void pack_qkv_wide_kernel(const float* __restrict__ w, bf16* __restrict__ dst, int D, float qscale) {
  const long long total = (long long)3 * D * D;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int row, col, mat;
    wide_src(idx, D, row, col, mat);
    dst[idx] = (bf16)(w[(size_t)row * D + col] * (mat == 0 ? qscale : 1.0f));
  }
}

}  // namespace vitpe

using namespace vitpe;

extern "C" int vitpe_fused_attention_wide_supported(int dtype, int N, int D, int HD) {
  return dtype == 1 && N == 65 && D == 192 && HD == 32;
}

extern "C" int vitpe_qkv_wide_pack_elems(int D) { return 3 * D * D; }

extern "C" int vitpe_pack_qkv_weights_wide(int dtype, const float* wqkv, void* packed, int D, int HD, hipStream_t stream) {
  VITPE_REQUIRE(wqkv && packed && dtype == 1 && HD == 32 && D > 0 && D % 32 == 0);
  const long long total = (long long)3 * D * D;
  const unsigned blocks = (unsigned)min((total + 255) / 256, (long long)2048);
  hipLaunchKernelGGL(pack_qkv_wide_kernel, dim3(blocks), dim3(256), 0, stream, wqkv, (bf16*)packed, D,
                     LOG2E / sqrtf((float)HD));
  VITPE_CHECK_LAUNCH();
}

template <int KM, bool MIXED = false>
static int launch_wide(const AttnArgs& a, hipStream_t s) {
  const dim3 grid((a.B + 1) / 2), block(768);
  if (a.ln_gamma != nullptr) hipLaunchKernelGGL((attn32_fwd_kernel<KM, true, MIXED>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((attn32_fwd_kernel<KM, false, MIXED>), grid, block, 0, s, a);
  VITPE_CHECK_LAUNCH();
}

// vitpe_fused_attention_fwd(_ln) on the wide kernel: wqkv_wide = vitpe_pack_qkv_weights_wide; gamma == NULL: x is already
// layer-normed (mean / rstd / xn_out ignored)
extern "C" int vitpe_fused_attention_fwd_wide(int dtype, const void* x, const float* gamma, const float* beta,
                                              const float* mean, const float* rstd, void* xn_out,
                                              const void* wqkv_wide, void* out, int B, int N, int D, int HD, int mode,
                                              const float* cos, const float* sin, const float* table,
                                              const float* coeff, int grid, int degree, int coeff_per_head,
                                              hipStream_t stream) {
  VITPE_REQUIRE(x && wqkv_wide && out && B >= 0);
  if (!vitpe_fused_attention_wide_supported(dtype, N, D, HD)) return (int)hipErrorNotSupported;
  if (gamma != nullptr) VITPE_REQUIRE(beta && mean && rstd);
  if (mode == PE_ROPE_AXIAL || mode == PE_ROPE_MIXED) VITPE_REQUIRE(cos && sin && grid * grid == N - 1);
  if (mode == PE_RELATIVE) VITPE_REQUIRE(table != nullptr);
  if (mode == PE_POLY) VITPE_REQUIRE(coeff && degree >= 0 && degree <= 7 && grid * grid == N - 1);
  VITPE_REQUIRE(mode >= PE_NONE && mode <= PE_ROPE_MIXED);
  if (B == 0) return 0;
  AttnArgs a{};
  a.xn = x; a.wqkv = wqkv_wide; a.out = out; a.cos = cos; a.sin = sin; a.table = table; a.coeff = coeff;
  a.ln_gamma = gamma; a.ln_beta = beta; a.ln_mean = mean; a.ln_rstd = rstd; a.xn_out = gamma ? xn_out : nullptr;
  a.B = B; a.N = N; a.mode = mode; a.grid = grid; a.degree = degree; a.coeff_per_head = coeff_per_head;
  a.scale = 1.0f / sqrtf((float)HD);
  switch (mode) {
    case PE_RELATIVE: return launch_wide<KM_RELATIVE>(a, stream);
    case PE_POLY: return launch_wide<KM_POLY>(a, stream);
    case PE_ROPE_AXIAL: return launch_wide<KM_ROPE>(a, stream);
    case PE_ROPE_MIXED: return launch_wide<KM_ROPE, true>(a, stream);
    default: return launch_wide<KM_PLAIN>(a, stream);
  }
}

// debug: phase census of the wide forward (rope-axial, no LayerNorm): census[(workgroup * 16 + wave) * 16 + slot] =
// s_memtime at 0 start, 1 staged, 2 barrier passed, 3 k-loop done, 4 operand fragments built, 5 patch queries done, 6 end;
// 8 = cycles inside the k-loop's barriers (incl. the LDS-DMA wait), 9 / 10 = s_memrealtime (100 MHz, chip-wide) at start / end
extern "C" int vitpe_debug_attn32_census(const void* xn, const void* wqkv_wide, void* out, const float* cos,
                                         const float* sin, int B, unsigned long long* census, int exp, hipStream_t stream) {
  VITPE_REQUIRE(xn && wqkv_wide && out && cos && sin && census && B > 0);
  AttnArgs a{};
  a.xn = xn; a.wqkv = wqkv_wide; a.out = out; a.cos = cos; a.sin = sin; a.B = B; a.N = 65; a.mode = PE_ROPE_AXIAL; a.grid = 8;
  a.scale = 0.17677669f; a.census = census;
  const dim3 grid((B + 1) / 2), block(768);
  if (exp == 1) hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true, 1>), grid, block, 0, stream, a);
  else if (exp == 2) hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true, 2>), grid, block, 0, stream, a);
  else if (exp == 3) hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true, 3>), grid, block, 0, stream, a);
  else if (exp == 4) hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true, 4>), grid, block, 0, stream, a);
  else if (exp == 8) hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true, 8>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn32_fwd_kernel<KM_ROPE, false, false, true>), grid, block, 0, stream, a);
  VITPE_CHECK_LAUNCH();
}
