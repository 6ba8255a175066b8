// Shared device-side building blocks for the vitpe HIP kernels (gfx950 / CDNA4 only).
//
// One tiling, two arithmetic types:
//   T = bf16  : v_mfma_f32_16x16x32_bf16, one MFMA per 32-deep K chunk (throughput mode)
//   T = float : v_mfma_f32_16x16x4_f32, eight MFMAs per 32-deep K chunk  (exact fp32, the
//               1e-4 parity gate against the reference's fp32 CPU path)
// Both use the same C/D map (col = lane&15, row = 4*(lane>>4)+reg) and the same operand
// convention below, so every kernel is written once and instantiated for both.
//
// Operand convention ("K32 chunk"): lane l = 16*g + c holds 8 elements t = 0..7 of
//   A[row c][k = 8g + t]   and   B[k = 8g + t][col c].
// For bf16 that is exactly the hardware map of mfma_f32_16x16x32_bf16.  For fp32 the t-th
// of eight 16x16x4 MFMAs contracts k' = g  <->  k = 8g + t; A and B use the same
// permutation of k, so the chunk sum is the same dot product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitpe {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define VITPE_DEV __device__ __forceinline__

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

// 16 bytes of padding per LDS row keeps 16 consecutive rows on distinct 16-B bank slots
// for ds_read_b128 fragment reads (row strides that are multiples of 64 B otherwise alias).
template <typename T> struct Pad { static constexpr int elems = 16 / (int)sizeof(T); };

VITPE_DEV float to_f32(float x) { return x; }
VITPE_DEV float to_f32(bf16 x) { return (float)x; }
template <typename T> VITPE_DEV T from_f32(float x);
template <> VITPE_DEV float from_f32<float>(float x) { return x; }
template <> VITPE_DEV bf16 from_f32<bf16>(float x) { return (bf16)x; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

// ---- MMA: c += A(16 x 32) * B(32 x 16) -------------------------------------------------
VITPE_DEV void mma(const Frag<bf16>& a, const Frag<bf16>& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
VITPE_DEV void mma(const Frag<float>& a, const Frag<float>& b, f32x4& c) {
#pragma unroll
  for (int t = 0; t < 8; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[t], b.v[t], c, 0, 0, 0);
}

// the fragment's value is FINAL here: keeps the compiler from sinking the conversions that produce it past later code
// (and carrying the fp32 sources instead -- DESIGN.md, round-2 log)
VITPE_DEV void pin_frag(Frag<bf16>& f) { asm volatile("" : "+v"(f.v)); }
VITPE_DEV void pin_frag(Frag<float>& f) {
#pragma unroll
  for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(f.v[t]));
}

template <typename T> VITPE_DEV Frag<T> zero_frag();
template <> VITPE_DEV Frag<bf16> zero_frag<bf16>() {
  Frag<bf16> f;
#pragma unroll
  for (int t = 0; t < 8; ++t) f.v[t] = (bf16)0.0f;
  return f;
}
template <> VITPE_DEV Frag<float> zero_frag<float>() {
  Frag<float> f;
#pragma unroll
  for (int t = 0; t < 8; ++t) f.v[t] = 0.0f;
  return f;
}

// ---- row fragment: 8 contiguous elements (16 B bf16 / 32 B fp32), 16-B aligned ----------
VITPE_DEV Frag<bf16> ld_frag(const bf16* p) {
  Frag<bf16> f;
  f.v = *reinterpret_cast<const bf16x8*>(p);
  return f;
}
VITPE_DEV Frag<float> ld_frag(const float* p) {
  Frag<float> f;
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) { f.v[t] = a[t]; f.v[4 + t] = b[t]; }
  return f;
}

// two separately addressed 16-B halves (fp32 fragments in XOR-swizzled LDS rows)
VITPE_DEV Frag<float> ld_frag2(const float* p0, const float* p1) {
  Frag<float> f;
  f32x4 a = *reinterpret_cast<const f32x4*>(p0);
  f32x4 b = *reinterpret_cast<const f32x4*>(p1);
#pragma unroll
  for (int t = 0; t < 4; ++t) { f.v[t] = a[t]; f.v[4 + t] = b[t]; }
  return f;
}
VITPE_DEV Frag<bf16> ld_frag2(const bf16* p0, const bf16*) { return ld_frag(p0); }  // never used for bf16

// ---- transposed fragment from a row-major LDS tile --------------------------------------
// Lane (c, g) receives elements t<4 : tile[(rb0 + t) * ld + c0 + c]
//                               t>=4: tile[(rb1 + t-4) * ld + c0 + c]
// rb0 / rb1 are the first rows of the lane-group's two 4-row blocks (they may depend on g).
// bf16: two ds_read_b64_tr_b16 (each 16-lane group reads a 4-row x 16-col block and gets it
// back column-major; lane 4q+p of the group supplies the address of row q, cols 4p..4p+3).
// Must be called with EXEC all ones (uniform control flow) -- the gather crosses lanes.
VITPE_DEV Frag<bf16> ld_frag_tr(const bf16* tile, int ld, int rb0, int rb1, int c0) {
  const int i = threadIdx.x & 15;
  const int q = i >> 2, p = i & 3;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const bf16* a0 = tile + (rb0 + q) * ld + c0 + 4 * p;
  const bf16* a1 = tile + (rb1 + q) * ld + c0 + 4 * p;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
  const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, both);
  return f;
}
VITPE_DEV Frag<float> ld_frag_tr(const float* tile, int ld, int rb0, int rb1, int c0) {
  const int c = threadIdx.x & 15;
  Frag<float> f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f.v[t] = tile[(rb0 + t) * ld + c0 + c];
    f.v[4 + t] = tile[(rb1 + t) * ld + c0 + c];
  }
  return f;
}

// ---- accumulator pair as the next MMA's operand -----------------------------------------
// Two 16x16 accumulator tiles X_lo, X_hi (rows 4g+r, col c on the lane) become the operand of
// a product that contracts over their ROW index: element t<4 <-> row 4g+t of X_lo, t>=4 <->
// row 4g+(t-4) of X_hi.  The partner operand must use ld_frag_tr(rb0 = base_lo + 4g,
// rb1 = base_hi + 4g) so both sides see the same k for every (g, t).
template <typename T> VITPE_DEV Frag<T> acc_to_frag(const f32x4& lo, const f32x4& hi);
template <> VITPE_DEV Frag<bf16> acc_to_frag<bf16>(const f32x4& lo, const f32x4& hi) {
  Frag<bf16> f;
#pragma unroll
  for (int t = 0; t < 4; ++t) { f.v[t] = (bf16)lo[t]; f.v[4 + t] = (bf16)hi[t]; }
  return f;
}
template <> VITPE_DEV Frag<float> acc_to_frag<float>(const f32x4& lo, const f32x4& hi) {
  Frag<float> f;
#pragma unroll
  for (int t = 0; t < 4; ++t) { f.v[t] = lo[t]; f.v[4 + t] = hi[t]; }
  return f;
}

// ---- packed 4-element stores / loads (8 B bf16, 16 B fp32) ------------------------------
VITPE_DEV void st4(bf16* p, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}
VITPE_DEV void st4(float* p, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  *reinterpret_cast<f32x4*>(p) = v;
}
VITPE_DEV f32x4 ld4(const bf16* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
VITPE_DEV f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// 16-byte chunk = CH<T>::n elements
template <typename T> struct CH { static constexpr int n = 16 / (int)sizeof(T); };
typedef __attribute__((ext_vector_type(4))) uint32_t Chunk16;  // first-class 16-B value (never an alloca)

template <typename T> VITPE_DEV void chunk_to_f32(const Chunk16& c, float* out);
template <> VITPE_DEV void chunk_to_f32<float>(const Chunk16& c, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(c[i]);
}
template <> VITPE_DEV void chunk_to_f32<bf16>(const Chunk16& c, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(c[i] << 16);
    out[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
  }
}
template <typename T> VITPE_DEV Chunk16 f32_to_chunk(const float* in);
template <> VITPE_DEV Chunk16 f32_to_chunk<float>(const float* in) {
  Chunk16 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(in[i]);
  return c;
}
template <> VITPE_DEV Chunk16 f32_to_chunk<bf16>(const float* in) {
  Chunk16 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x2 v;
    v[0] = (bf16)in[2 * i];
    v[1] = (bf16)in[2 * i + 1];
    c[i] = *reinterpret_cast<uint32_t*>(&v);
  }
  return c;
}

// ---- wavefront (64-lane) reductions ------------------------------------------------------
VITPE_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// across the four 16-lane groups (lanes c, c+16, c+32, c+48)
VITPE_DEV float xgroup_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
VITPE_DEV float xgroup_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
// within a 16-lane group
VITPE_DEV float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
VITPE_DEV float group16_max(float v) {
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact-erf GELU (nn.GELU default, reference vit.py:111) and its derivative.  erf through
// Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far inside the 1e-4 parity gate): one rcp, one
// exp2 and a degree-5 Horner polynomial instead of libm erff's ~35 instructions -- these sit in GEMM
// epilogues that evaluate them 36 times per lane per tile.  The branch on the sign avoids the
// 1 + erf(x) cancellation for negative arguments.  cdf = Phi(u), e = exp(-u^2/2).
VITPE_DEV float gelu_cdf(float u, float& e) {
  const float ax = fabsf(u) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float half = 0.5f * p * e;  // 0.5 * erfc(|x|)
  return u >= 0.f ? 1.0f - half : half;
}
VITPE_DEV float gelu_erf(float u) {
  float e;
  return u * gelu_cdf(u, e);
}
VITPE_DEV float gelu_erf_grad(float u) {
  float e;
  const float cdf = gelu_cdf(u, e);
  return fmaf(u * 0.39894228040143267794f, e, cdf);
}

// The same on element PAIRS: ext-vector float2 arithmetic compiles to v_pk_mul_f32 / v_pk_fma_f32 (two lanes'
// worth of fp32 per issue slot), 23 instead of 38 VALU instructions per pair -- the GELU sits in epilogues that are
// VALU-bound (fused MLP kernels: ~1/3 of their time).
VITPE_DEV f32x2 gelu_cdf2(f32x2 u, f32x2& e) {
  const f32x2 ax = {fabsf(u[0]) * 0.70710678118654752440f, fabsf(u[1]) * 0.70710678118654752440f};
  const f32x2 d = ax * 0.3275911f + (f32x2){1.0f, 1.0f};
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = t * 1.061405429f + (f32x2){-1.453152027f, -1.453152027f};
  p = t * p + (f32x2){1.421413741f, 1.421413741f};
  p = t * p + (f32x2){-0.284496736f, -0.284496736f};
  p = t * p + (f32x2){0.254829592f, 0.254829592f};
  p = p * t;
  const f32x2 a2 = ax * ax * -1.4426950408889634f;
  e = (f32x2){__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
  const f32x2 half = p * e * 0.5f;  // 0.5 * erfc(|x|)
  f32x2 r;
  r[0] = u[0] >= 0.f ? 1.0f - half[0] : half[0];
  r[1] = u[1] >= 0.f ? 1.0f - half[1] : half[1];
  return r;
}
// v[0..7] <- gelu(v[0..7])
VITPE_DEV void gelu_erf_x8(float* v) {
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    f32x2 e;
    const f32x2 u = {v[t], v[t + 1]};
    const f32x2 r = u * gelu_cdf2(u, e);
    v[t] = r[0];
    v[t + 1] = r[1];
  }
}
// v[0..7] <- v[0..7] * gelu'(u[0..7])
VITPE_DEV void gelu_erf_grad_mul_x8(float* v, const float* uv) {
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    f32x2 e;
    const f32x2 u = {uv[t], uv[t + 1]};
    const f32x2 cdf = gelu_cdf2(u, e);
    const f32x2 gr = u * 0.39894228040143267794f * e + cdf;
    v[t] *= gr[0];
    v[t + 1] *= gr[1];
  }
}

// positional-encoding modes (include/vitpe.h VITPE_PE_*)
enum { PE_NONE = 0, PE_ABSOLUTE = 1, PE_RELATIVE = 2, PE_POLY = 3, PE_ROPE_AXIAL = 4, PE_ROPE_MIXED = 5 };

}  // namespace vitpe

// error plumbing for the C-ABI: every entry point returns hipError_t as int, never throws
#define VITPE_CHECK_LAUNCH() return (int)hipGetLastError()
#define VITPE_REQUIRE(cond) do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)
