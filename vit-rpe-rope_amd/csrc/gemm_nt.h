// Argument block and epilogue codes shared by the GEMM kernels of gemm.hip and gemm2d.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace vitpe {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_PATCH = 3, EPI_GELU_BWD = 4, EPI_LN_BWD = 5 };

struct GemmNTArgs {
  const void* A;      // [M,K] T
  const void* W;      // [N,K] T
  void* C;            // [M,N] T   (EPI_PATCH: [B*Ntok, N])
  const float* bias;  // [N] fp32 or null
  const void* R;      // EPI_BIAS_RESID: residual [M,N] T
  void* U;            // EPI_BIAS_GELU: pre-activation out [M,N] T ; EPI_GELU_BWD: pre-activation in
  const float* ape;   // EPI_PATCH: absolute PE rows [P,N] fp32 or null
  const float* cls;   // EPI_PATCH: class token [N] fp32
  int M, N, K;
  int P, Ntok;        // EPI_PATCH: patches per image, tokens per image (P+1)
};

// gemm2d.hip: big-tile bf16 kernel for the many-row / wide-weight shapes (ViT-B/16 geometry)
bool gemm2d_takes(int dtype, int epi, int M, int N, int K);
int gemm2d_launch(int epi, const GemmNTArgs& a, hipStream_t s);

}  // namespace vitpe
