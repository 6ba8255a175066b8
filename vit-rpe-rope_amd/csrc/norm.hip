// LayerNorm forward / backward (reference nn.LayerNorm(d), eps 1e-5: models/vit.py:113,116,210).
// HBM-bound: one wavefront per token row, 16-byte loads, two-pass statistics in registers,
// fp32 math.  Backward also folds in the residual-branch gradient (dx = dres + LN'(dy)) so the
// residual add of vit.py:122,124 costs no extra pass, and reduces dgamma/dbeta through
// per-workgroup partial rows summed in a fixed order (deterministic).
#include "common.h"

namespace vitpe {

constexpr int LN_MAXC = 4;  // 16-B chunks cached per lane: D <= 64*4*8 (bf16) / 64*4*4 (fp32)

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int M, int D, float eps) {
  constexpr int CHN = CH<T>::n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D / CHN;
  const float invD = 1.0f / (float)D;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float v[LN_MAXC][CHN];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + (size_t)row * D + ch * CHN), v[i]);
#pragma unroll
        for (int t = 0; t < CHN; ++t) s += v[i][t];
      }
    }
    const float mean = wave_sum(s) * invD;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int t = 0; t < CHN; ++t) { const float d = v[i][t] - mean; s2 += d * d; }
      }
    }
    const float var = wave_sum(s2) * invD;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[CHN];
#pragma unroll
        for (int t = 0; t < CHN; ++t)
          o[t] = (v[i][t] - mean) * rstd * gamma[ch * CHN + t] + beta[ch * CHN + t];
        if (y != nullptr) *reinterpret_cast<Chunk16*>(y + (size_t)row * D + ch * CHN) = f32_to_chunk<T>(o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// dx[row] = (dres ? dres[row] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy  (per-workgroup partial sums, then fp32 atomics)
// MAXC = 16-B chunks cached per lane, NW = waves per workgroup.  <4, 4> is the general shape; <2, 16> (D <= 1024 in bf16)
// halves the register arrays so that 16 waves fit a CU and puts 4 rows per SIMD in flight instead of one -- a row is one
// dependent load -> reduce -> store chain, and with 1 024 waves on the chip the d = 768 launch ran at 1.8 TB/s.
template <typename T, int MAXC, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int D) {
  constexpr int CHN = CH<T>::n;
  extern __shared__ __attribute__((aligned(16))) float sred[];  // [NW waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = D / CHN;
  const float invD = 1.0f / (float)D;
  float ag[MAXC][CHN], ab[MAXC][CHN], gm[MAXC][CHN];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int t = 0; t < CHN; ++t) {
      ag[i][t] = 0.f; ab[i][t] = 0.f;
      gm[i][t] = (ch < nch) ? gamma[ch * CHN + t] : 0.f;
    }
  }
  for (int row = blockIdx.x * NW + wave; row < M; row += gridDim.x * NW) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[MAXC][CHN], g[MAXC][CHN];
    float s1 = 0.f, s2 = 0.f;
    // the residual rows are requested WITH x and dy (packed, 4 registers per chunk): asked for behind the row reductions
    // they were a second exposed memory latency per row
    Chunk16 rres[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      rres[i] = (Chunk16){0u, 0u, 0u, 0u};
      if (dres != nullptr && ch < nch) rres[i] = *reinterpret_cast<const Chunk16*>(dres + (size_t)row * D + ch * CHN);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float xv[CHN], dv[CHN];
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(x + (size_t)row * D + ch * CHN), xv);
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(dy + (size_t)row * D + ch * CHN), dv);
#pragma unroll
        for (int t = 0; t < CHN; ++t) {
          xh[i][t] = (xv[t] - mean) * rstd;
          g[i][t] = dv[t] * gm[i][t];
          s1 += g[i][t];
          s2 += g[i][t] * xh[i][t];
          ag[i][t] += dv[t] * xh[i][t];
          ab[i][t] += dv[t];
        }
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[CHN];
#pragma unroll
        for (int t = 0; t < CHN; ++t) o[t] = rstd * (g[i][t] - s1 - xh[i][t] * s2);
        if (dres != nullptr) {
          float rv[CHN];
          chunk_to_f32<T>(rres[i], rv);
#pragma unroll
          for (int t = 0; t < CHN; ++t) o[t] += rv[t];
        }
        *reinterpret_cast<Chunk16*>(dx + (size_t)row * D + ch * CHN) = f32_to_chunk<T>(o);
      }
    }
  }
  // cross-wave reduction, fixed order; NS = min(NW, 8) slots of [2][D] (16 waves fold pairwise first: 16 slots of d = 768
  // would be 96 KB of dynamic LDS)
  constexpr int NS = NW < 8 ? NW : 8;
  auto park = [&](int slot) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int t = 0; t < CHN; ++t) {
          sred[(slot * 2 + 0) * D + ch * CHN + t] = ag[i][t];
          sred[(slot * 2 + 1) * D + ch * CHN + t] = ab[i][t];
        }
      }
    }
  };
  if (NW > NS) {
    if (wave >= NS) park(wave - NS);
    __syncthreads();
    if (wave < NS) {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
          for (int t = 0; t < CHN; ++t) {
            ag[i][t] += sred[(wave * 2 + 0) * D + ch * CHN + t];
            ab[i][t] += sred[(wave * 2 + 1) * D + ch * CHN + t];
          }
        }
      }
    }
    __syncthreads();
  }
  if (wave < NS) park(wave);
  __syncthreads();
  // 2*D atomics per workgroup onto 2*D addresses: <= 256 workgroups, a few microseconds
  for (int i = threadIdx.x; i < 2 * D; i += NW * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += sred[2 * D * w + i];
    atomicAdd((i < D ? dgamma + i : dbeta + (i - D)), s);
  }
}

// ---- bf16, D <= 256: a row is <= 32 chunks, so every wave works on TWO rows at once (one per
// 32-lane half) and a block keeps 8 rows in flight; reductions are 5 shuffles inside the half.
__global__ __launch_bounds__(256) void ln_fwd32_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16* __restrict__ y,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int M, int D, float eps) {
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;  // 8 half-waves per block
  const int nch = D / 8;
  const float invD = 1.0f / (float)D;
  const bool act = lane < nch;
  float gm[8], bt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { gm[t] = act ? gamma[lane * 8 + t] : 0.f; bt[t] = act ? beta[lane * 8 + t] : 0.f; }
  for (int row = blockIdx.x * 8 + sub; row < M; row += gridDim.x * 8) {
    float v[8];
    float s = 0.f;
    if (act) {
      chunk_to_f32<bf16>(*reinterpret_cast<const Chunk16*>(x + (size_t)row * D + lane * 8), v);
#pragma unroll
      for (int t = 0; t < 8; ++t) s += v[t];
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * invD;
    float s2 = 0.f;
    if (act) {
#pragma unroll
      for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; s2 += d * d; }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    const float rstd = 1.0f / sqrtf(s2 * invD + eps);
    if (act) {
      float o8[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o8[t] = (v[t] - mean) * rstd * gm[t] + bt[t];
      if (y != nullptr) *reinterpret_cast<Chunk16*>(y + (size_t)row * D + lane * 8) = f32_to_chunk<bf16>(o8);
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// 1024 threads = 32 half-waves per block: one block per CU keeps 64 rows in flight while the
// dgamma/dbeta atomics stay at one per column and block
__global__ __launch_bounds__(1024) void ln_bwd32_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       const float* __restrict__ gamma, const bf16* __restrict__ dres,
                                                       bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int M, int D) {
  __shared__ float sred[32][2][256];
  const int lane = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int nch = D / 8;
  const float invD = 1.0f / (float)D;
  const bool act = lane < nch;
  float gm[8], ag[8], ab[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { gm[t] = act ? gamma[lane * 8 + t] : 0.f; ag[t] = 0.f; ab[t] = 0.f; }
  // two rows per half-wave in flight: all loads of both rows are issued before the first reduction
  const int stride = gridDim.x * 32;
  for (int row0 = blockIdx.x * 32 + sub; row0 < M; row0 += 2 * stride) {
    float xh[2][8], g8[2][8], rv[2][8], dv[2][8];
    float mean[2], rstd[2], s1[2], s2[2];
    bool ok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = row0 + k * stride;
      ok[k] = act && row < M;
      mean[k] = rstd[k] = 0.f;
      if (row < M) { mean[k] = mean_in[row]; rstd[k] = rstd_in[row]; }
      if (ok[k]) {
        chunk_to_f32<bf16>(*reinterpret_cast<const Chunk16*>(x + (size_t)row * D + lane * 8), xh[k]);
        chunk_to_f32<bf16>(*reinterpret_cast<const Chunk16*>(dy + (size_t)row * D + lane * 8), dv[k]);
        if (dres != nullptr) chunk_to_f32<bf16>(*reinterpret_cast<const Chunk16*>(dres + (size_t)row * D + lane * 8), rv[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      s1[k] = s2[k] = 0.f;
      if (ok[k]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          xh[k][t] = (xh[k][t] - mean[k]) * rstd[k];
          g8[k][t] = dv[k][t] * gm[t];
          s1[k] += g8[k][t];
          s2[k] += g8[k][t] * xh[k][t];
          ag[t] += dv[k][t] * xh[k][t];
          ab[t] += dv[k][t];
        }
      }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
      s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
      s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (ok[k]) {
        const int row = row0 + k * stride;
        const float m1 = s1[k] * invD, m2 = s2[k] * invD;
        float o8[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) o8[t] = rstd[k] * (g8[k][t] - m1 - xh[k][t] * m2) + (dres != nullptr ? rv[k][t] : 0.f);
        *reinterpret_cast<Chunk16*>(dx + (size_t)row * D + lane * 8) = f32_to_chunk<bf16>(o8);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) { sred[sub][0][lane * 8 + t] = ag[t]; sred[sub][1][lane * 8 + t] = ab[t]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 1024) {
    const int which = i / D, d = i % D;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += sred[k][which][d];
    atomicAdd((which == 0 ? dgamma : dbeta) + d, t);
  }
}

// dst0[i] += sum_p partial[p][i] (i < len0) ; dst1[i-len0] += ... (len0 <= i < len0+len1)
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int len0, int len1,
                                       float* __restrict__ dst0, float* __restrict__ dst1) {
  // block = 32 columns x 8 partial-groups; LDS tree over the groups (fixed order)
  __shared__ float red[8][33];
  const int len = len0 + len1;
  const int i = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  float s = 0.f;
  if (i < len)
    for (int p = grp; p < nparts; p += 8) s += partial[(size_t)p * len + i];
  red[grp][threadIdx.x & 31] = s;
  __syncthreads();
  if (grp == 0 && i < len) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    if (i < len0) dst0[i] += t; else dst1[i - len0] += t;
  }
}

}  // namespace vitpe

using namespace vitpe;

extern "C" int vitpe_reduce_partials(const float* partial, int nparts, int len0, int len1, float* dst0,
                                     float* dst1, hipStream_t stream) {
  VITPE_REQUIRE(partial && dst0 && nparts >= 0 && len0 >= 0 && len1 >= 0 && (len1 == 0 || dst1));
  const int len = len0 + len1;
  if (len == 0) return 0;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + 31) / 32), dim3(256), 0, stream, partial, nparts,
                     len0, len1, dst0, dst1);
  VITPE_CHECK_LAUNCH();
}

static inline bool ln_dims_ok(int dtype, int D) {
  const int chn = dtype == 1 ? 8 : 4;
  return D > 0 && D % chn == 0 && D / chn <= 64 * LN_MAXC;
}

extern "C" int vitpe_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int M, int D, float eps, hipStream_t stream) {
  VITPE_REQUIRE(x && gamma && beta && M >= 0);   // y == NULL: statistics only
  VITPE_REQUIRE((y != nullptr) || (mean && rstd));
  VITPE_REQUIRE((dtype == 0 || dtype == 1) && ln_dims_ok(dtype, D));
  if (M == 0) return 0;
  const int blocks = min((M + 3) / 4, 2048);
  if (dtype == 1 && D <= 256) {
    hipLaunchKernelGGL(ln_fwd32_kernel, dim3(min((M + 7) / 8, 1024)), dim3(256), 0, stream, (const bf16*)x, gamma, beta,
                       (bf16*)y, mean, rstd, M, D, eps);
    VITPE_CHECK_LAUNCH();
  }
  if (dtype == 1)
    hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, gamma, beta,
                       (bf16*)y, mean, rstd, M, D, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, gamma, beta,
                       (float*)y, mean, rstd, M, D, eps);
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_layernorm_bwd_blocks(int M) { return min((M + 3) / 4, 256); }

// workspace: vitpe_layernorm_bwd_blocks(M) * 2 * D floats
extern "C" int vitpe_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                                   const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                   float* workspace, int M, int D, hipStream_t stream) {
  VITPE_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && M >= 0);
  (void)workspace;  // kept in the ABI; no longer needed
  VITPE_REQUIRE((dtype == 0 || dtype == 1) && ln_dims_ok(dtype, D));
  if (M == 0) return 0;
  const int blocks = vitpe_layernorm_bwd_blocks(M);
  const size_t shm = (size_t)4 * 2 * D * sizeof(float);
  if (dtype == 1 && D <= 256) {
    hipLaunchKernelGGL(ln_bwd32_kernel, dim3(min((M + 63) / 64, 256)), dim3(1024), 0, stream, (const bf16*)dy, (const bf16*)x,
                       mean, rstd, gamma, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, M, D);
    VITPE_CHECK_LAUNCH();
  }
  if (dtype == 1 && D <= 1024) {
    const int blocks16 = min((M + 15) / 16, 256);
    hipLaunchKernelGGL((ln_bwd_kernel<bf16, 2, 16>), dim3(blocks16), dim3(1024), (size_t)8 * 2 * D * sizeof(float), stream,
                       (const bf16*)dy, (const bf16*)x, mean, rstd, gamma, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, M, D);
  } else if (dtype == 1) {
    hipLaunchKernelGGL((ln_bwd_kernel<bf16, LN_MAXC, 4>), dim3(blocks), dim3(256), shm, stream, (const bf16*)dy, (const bf16*)x,
                       mean, rstd, gamma, (const bf16*)dres, (bf16*)dx, dgamma, dbeta, M, D);
  } else {
    hipLaunchKernelGGL((ln_bwd_kernel<float, LN_MAXC, 4>), dim3(blocks), dim3(256), shm, stream, (const float*)dy,
                       (const float*)x, mean, rstd, gamma, (const float*)dres, (float*)dx, dgamma, dbeta, M, D);
  }
  VITPE_CHECK_LAUNCH();
}
