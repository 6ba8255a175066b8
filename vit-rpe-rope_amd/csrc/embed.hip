// Fused patch embedding (reference models/vit.py:164,245-258 + positional_encoding.py:37-40 + the first block's
// norm1 statistics, vit.py:113):
//
//   image -> unfold (Conv2d k = s = p  ==  patch matrix x W^T) -> + bias -> + absolute PE -> class token in row 0
//         -> tokens [B, P+1, D]  (+ LayerNorm statistics of every token row for the first attention kernel)
//
// in ONE kernel for the small-K geometries (K = C p^2 <= 64, P <= 64, D <= 256: the CIFAR / MNIST shapes), one workgroup
// (4 waves) per image.  The unfold is the A-operand staging: the image's pixels go global -> registers -> LDS in patch
// order (from fp32 images, or gathered + ToTensor + Normalize'd from the resident uint8 dataset, train.py:69-92); the
// [P, K] patch matrix never makes an HBM round trip on the way to the MFMAs.  It is still written out once (3 MB at
// batch 512) because the patch-embed WEIGHT gradient needs it as its X operand in the backward pass.
// The product runs on the matrix core (4 token tiles x D/16 feature tiles x K/32 chunks); bias / PE add, the row
// statistics and the stores happen on the accumulators (a lane owns four consecutive features of one token per tile).
// Replaces three launches (vitpe_unfold[_u8] + vitpe_gemm_nt(EPI_PATCH) + vitpe_layernorm_fwd statistics).
#include "common.h"

namespace vitpe {

struct EmbedArgs {
  const float* img;            // [B,C,S,S] fp32, or null when data is set
  const unsigned char* data;   // resident uint8 dataset [Ndata,C,S,S] or null
  const long long* index;      // [B] sample indices into data (null: record b)
  const float* nmean;          // [C] Normalize constants (uint8 path)
  const float* nstd;
  const void* W;               // patch_embed.weight [D, K] T (Conv2d weight flattened: k = c p^2 + ky p + kx)
  const float* bias;           // [D]
  const float* cls;            // [D]
  const float* ape;            // [P, D] rows of the absolute PE table, or null
  void* tokens;                // [B, P+1, D] T
  void* patches;               // [B*P, K] T (for the weight gradient), nullable
  float* mean;                 // [B*(P+1)] LayerNorm statistics of the token rows, nullable (both or neither)
  float* rstd;
  int B, C, S, p, D;
  float eps;
};

constexpr int EMB_KP = 64;     // padded K
constexpr int EMB_MP = 64;     // padded patches per image
constexpr int EMB_DMAX = 256;

// One workgroup = one image = 4 waves, wave w owns the 16-token tile w through ALL feature tiles.  Swapped MFMA
// orientation (A = weight rows, B = patch rows): acc[nt][r] = out[token 16w + c][feature 16nt + 4g + r], so a lane holds
// four consecutive features of ONE token per tile: the row statistics are an in-lane sum over the tiles plus a
// cross-group (g) reduction with two permlane swaps, and the stores are 8 B per lane -- no fp32 staging of the result.
template <typename T>
__global__ __launch_bounds__(256) void patch_embed_kernel(EmbedArgs a) {
  constexpr int LDA = EMB_KP + 2 * Pad<T>::elems;    // operand rows: 10 (bf16) slots, == 2 mod 4: conflict-free fragment reads
  constexpr int CHN = CH<T>::n, NTMAX = EMB_DMAX / 16, KSM = EMB_KP / 32;
  __shared__ __attribute__((aligned(16))) T sA[EMB_MP * LDA];
  __shared__ __attribute__((aligned(16))) T sW[EMB_DMAX * LDA];

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int C = a.C, S = a.S, p = a.p, D = a.D;
  const int G = S / p, P = G * G, K = C * p * p;
  const T* Wg = reinterpret_cast<const T*>(a.W);
  const Chunk16 zero = {0u, 0u, 0u, 0u};

  // ---- stage W rows (16-B chunks; the K padding zeroed) and the image's patches (the unfold) ---------------------------
  {
    constexpr int CPR = EMB_KP / CHN;              // chunks per padded row
    const int cpr = K / CHN;                       // real chunks per row (K % CHN == 0: support check)
    for (int q = tid; q < D * CPR; q += 256) {
      const int row = q / CPR, cc = q % CPR;
      *reinterpret_cast<Chunk16*>(sW + row * LDA + cc * CHN) =
          cc < cpr ? *reinterpret_cast<const Chunk16*>(Wg + (size_t)row * K + cc * CHN) : zero;
    }
    for (int q = tid; q < EMB_MP * CPR; q += 256) {   // K padding of every patch row, all of the rows past P
      const int row = q / CPR, cc = q % CPR;
      if (cc >= cpr || row >= P) *reinterpret_cast<Chunk16*>(sA + row * LDA + cc * CHN) = zero;
    }
  }
  const long long rec = a.data != nullptr ? (a.index != nullptr ? a.index[b] : (long long)b) : 0;
  T* pout = a.patches != nullptr ? reinterpret_cast<T*>(a.patches) + (size_t)b * P * K : nullptr;
  for (int q = tid; q < P * C * p; q += 256) {           // (patch n, channel, row ky): p contiguous pixels
    const int ky = q % p, ch = (q / p) % C, n = q / (p * C);
    const int gy = n / G, gx = n % G;
    const size_t pix = ((size_t)ch * S + gy * p + ky) * S + gx * p;
    const int k0 = ch * p * p + ky * p;
    if (p == 4) {   // the CIFAR / MNIST patch: one 16-B (fp32) or 4-B (uint8) load, one 8-B / 16-B store each way
      float v[4];
      if (a.data != nullptr) {
        const uchar4 u = *reinterpret_cast<const uchar4*>(a.data + (size_t)rec * C * S * S + pix);
        const float m = a.nmean[ch], sd = a.nstd[ch];   // ToTensor then Normalize, IEEE division (as vitpe_unfold_u8)
        v[0] = ((float)u.x / 255.0f - m) / sd; v[1] = ((float)u.y / 255.0f - m) / sd;
        v[2] = ((float)u.z / 255.0f - m) / sd; v[3] = ((float)u.w / 255.0f - m) / sd;
      } else {
        const f32x4 f = *reinterpret_cast<const f32x4*>(a.img + (size_t)b * C * S * S + pix);
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
      }
      st4(sA + n * LDA + k0, v[0], v[1], v[2], v[3]);
      if (pout != nullptr) st4(pout + (size_t)n * K + k0, v[0], v[1], v[2], v[3]);
    } else {
      for (int kx = 0; kx < p; ++kx) {
        float v;
        if (a.data != nullptr)
          v = ((float)a.data[(size_t)rec * C * S * S + pix + kx] / 255.0f - a.nmean[ch]) / a.nstd[ch];
        else
          v = a.img[(size_t)b * C * S * S + pix + kx];
        const T t = from_f32<T>(v);
        sA[n * LDA + k0 + kx] = t;
        if (pout != nullptr) pout[(size_t)n * K + k0 + kx] = t;
      }
    }
  }
  __syncthreads();

  T* tok = reinterpret_cast<T*>(a.tokens) + (size_t)b * (P + 1) * D;
  const float invD = 1.0f / (float)D;
  const int NTL = D / 16, KS = (K + 31) / 32;
  if (16 * wave < P) {
    const int n = 16 * wave + c;                      // this lane's patch (token n + 1)
    Frag<T> fa[KSM];
#pragma unroll
    for (int ks = 0; ks < KSM; ++ks) fa[ks] = ld_frag(sA + n * LDA + 32 * ks + 8 * g);
    float s1 = 0.f, s2 = 0.f;
    f32x4 keep[NTMAX];
#pragma unroll
    for (int nt = 0; nt < NTMAX; ++nt) {
      if (nt < NTL) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks)
          if (ks < KS) mma(ld_frag(sW + (16 * nt + c) * LDA + 32 * ks + 8 * g), fa[ks], acc);
        const int gn = 16 * nt + 4 * g;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + gn);
        acc += bv;
        if (a.ape != nullptr && n < P) acc += *reinterpret_cast<const f32x4*>(a.ape + (size_t)n * D + gn);
        // statistics of the values as stored
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] = to_f32(from_f32<T>(acc[r])); s1 += acc[r]; }
        keep[nt] = acc;
        if (n < P) st4(tok + (size_t)(n + 1) * D + gn, acc[0], acc[1], acc[2], acc[3]);
      }
    }
    if (a.mean != nullptr) {   // two-pass variance on the register-resident row (as the LayerNorm kernels)
      auto xg = [](float v) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        auto q2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
      };
      const float mean = xg(s1) * invD;
#pragma unroll
      for (int nt = 0; nt < NTMAX; ++nt)
        if (nt < NTL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float d = keep[nt][r] - mean; s2 += d * d; }
        }
      const float var = xg(s2) * invD;
      if (g == 0 && n < P) {
        a.mean[(size_t)b * (P + 1) + n + 1] = mean;
        a.rstd[(size_t)b * (P + 1) + n + 1] = 1.0f / sqrtf(var + a.eps);
      }
    }
  }
  // ---- class-token row (reference vit.py:253-254; APE skips it): the last wave, 8 columns per lane -----------------------
  if (wave == 3) {
    const int gn = lane * 8;
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = 0.f;
    if (gn < D) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = a.cls[gn + t];
#pragma unroll
      for (int h = 0; h < 8 / CHN; ++h) {
        const Chunk16 ch = f32_to_chunk<T>(v + h * CHN);
        *reinterpret_cast<Chunk16*>(tok + gn + h * CHN) = ch;
        chunk_to_f32<T>(ch, v + h * CHN);
      }
    }
    if (a.mean != nullptr) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) s += v[t];
      const float mean = wave_sum(s) * invD;
      float sq = 0.f;
      if (gn < D) {
#pragma unroll
        for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; sq += d * d; }
      }
      sq = wave_sum(sq);
      if (lane == 0) {
        a.mean[(size_t)b * (P + 1)] = mean;
        a.rstd[(size_t)b * (P + 1)] = 1.0f / sqrtf(sq * invD + a.eps);
      }
    }
  }
}

}  // namespace vitpe

using namespace vitpe;

extern "C" int vitpe_patch_embed_supported(int dtype, int C, int S, int p, int D) {
  if (!(dtype == 0 || dtype == 1) || C < 1 || p < 1 || S < p || S % p != 0) return 0;
  const int G = S / p, P = G * G, K = C * p * p;
  return K <= EMB_KP && K % (dtype == 1 ? 8 : 4) == 0 && P <= EMB_MP && D >= 16 && D <= EMB_DMAX && D % 16 == 0;
}

// img XOR data: fp32 images [B,C,S,S], or the resident uint8 dataset + sample indices (as vitpe_unfold_u8).
extern "C" int vitpe_patch_embed(int dtype, const float* img, const unsigned char* data, const long long* index,
                                 const float* nmean, const float* nstd, const void* W, const float* bias, const float* cls,
                                 const float* ape, void* tokens, void* patches, float* mean, float* rstd, int B, int C, int S,
                                 int p, int D, float eps, hipStream_t stream) {
  VITPE_REQUIRE(W && bias && cls && tokens && B >= 0);
  VITPE_REQUIRE((img != nullptr) != (data != nullptr));
  VITPE_REQUIRE(data == nullptr || (nmean && nstd));
  VITPE_REQUIRE((mean == nullptr) == (rstd == nullptr));
  if (!vitpe_patch_embed_supported(dtype, C, S, p, D)) return (int)hipErrorNotSupported;
  if (B == 0) return 0;
  EmbedArgs a{};
  a.img = img; a.data = data; a.index = index; a.nmean = nmean; a.nstd = nstd; a.W = W; a.bias = bias; a.cls = cls; a.ape = ape;
  a.tokens = tokens; a.patches = patches; a.mean = mean; a.rstd = rstd; a.B = B; a.C = C; a.S = S; a.p = p; a.D = D; a.eps = eps;
  if (dtype == 1) hipLaunchKernelGGL(patch_embed_kernel<bf16>, dim3(B), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(patch_embed_kernel<float>, dim3(B), dim3(256), 0, stream, a);
  VITPE_CHECK_LAUNCH();
}
