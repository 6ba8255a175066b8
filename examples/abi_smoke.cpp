// Stand-alone caller of the C ABI (include/vitpe.h) -- no PyTorch, no Python: device memory from hipMalloc, one
// transformer-block forward at the CIFAR geometry in bf16 through the same entry points the Python host binds:
//   weight packing -> LayerNorm statistics -> fused attention, 32x32-tile kernel (LayerNorm folded in) -> block tail
//   (proj + residual + LN2 + MLP + residual) -- the kernels the training step runs.
// Inputs are closed-form (sin-based) so the result is reproducible; prints a checksum of the block output that
// tests/test_c_abi_harness_gpu.py compares with the same computation issued through the Python host.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/abi_smoke.cpp -Lvit-rpe-rope_amd/lib -lvitpe -Wl,-rpath,$PWD/vit-rpe-rope_amd/lib -o abi_smoke
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "vitpe.h"

#define CK(x) do { int e_ = (int)(x); if (e_ != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, e_, __LINE__); return 1; } } while (0)

static uint16_t f2bf(float f) {  // round-to-nearest-even bf16
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float wave(size_t i, float salt, float amp) { return amp * sinf(0.37f * (float)i + salt); }

template <typename T> static T* dev(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
template <typename T> static T* devz(size_t n) {
  T* d = nullptr;
  if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return nullptr;
  hipMemset(d, 0, n * sizeof(T));
  return d;
}

int main() {
  const int B = 4, N = 65, D = 192, H = 6, HD = 32, HID = 768, G = 8, M = B * N;
  if (vitpe_abi_version() != 4) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  if (!vitpe_fused_attention_supported(VITPE_BF16, N, D, HD) || !vitpe_fused_attention_wide_supported(VITPE_BF16, N, D, HD) ||
      !vitpe_block_tail2_supported(VITPE_BF16, D, HID)) {
    fprintf(stderr, "geometry not supported\n");
    return 1;
  }
  std::vector<uint16_t> x((size_t)M * D);
  std::vector<float> wp((size_t)D * D), w1((size_t)HID * D), w2((size_t)D * HID);   // fp32 masters: the pack kernels round
  std::vector<float> wqkv((size_t)3 * D * D), g1(D), b1(D), g2(D), b2(D), bp(D), bf1(HID), bf2(D), invf(HD / 4);
  for (size_t i = 0; i < x.size(); ++i) x[i] = f2bf(wave(i, 0.1f, 1.0f));
  for (size_t i = 0; i < wqkv.size(); ++i) wqkv[i] = wave(i, 0.7f, 0.08f);
  for (size_t i = 0; i < wp.size(); ++i) wp[i] = wave(i, 1.3f, 0.07f);
  for (size_t i = 0; i < w1.size(); ++i) w1[i] = wave(i, 2.1f, 0.07f);
  for (size_t i = 0; i < w2.size(); ++i) w2[i] = wave(i, 2.9f, 0.04f);
  for (int i = 0; i < D; ++i) { g1[i] = 1.0f + wave(i, 3.3f, 0.1f); b1[i] = wave(i, 3.9f, 0.1f); g2[i] = 1.0f + wave(i, 4.4f, 0.1f);
                                b2[i] = wave(i, 5.0f, 0.1f); bp[i] = wave(i, 5.5f, 0.05f); bf2[i] = wave(i, 6.1f, 0.05f); }
  for (int i = 0; i < HID; ++i) bf1[i] = wave(i, 6.6f, 0.05f);
  for (int i = 0; i < HD / 4; ++i) invf[i] = 1.0f / powf(100.0f, (float)(4 * i) / (float)HD);   // rope_axial inv_freq, theta = 100

  hipStream_t st;
  CK(hipStreamCreate(&st));
  uint16_t* dx = dev(x);
  float *dwp = dev(wp), *dw1 = dev(w1), *dw2 = dev(w2);
  float *dwqkv = dev(wqkv), *dg1 = dev(g1), *db1 = dev(b1), *dg2 = dev(g2), *db2 = dev(b2), *dbp = dev(bp), *dbf1 = dev(bf1),
        *dbf2 = dev(bf2), *dinv = dev(invf);
  uint16_t* dpack = devz<uint16_t>((size_t)vitpe_qkv_wide_pack_elems(D));
  uint16_t *pwp = devz<uint16_t>(wp.size()), *pw1 = devz<uint16_t>(w1.size()), *pw2 = devz<uint16_t>(w2.size());
  float *m1 = devz<float>(M), *r1 = devz<float>(M), *m2 = devz<float>(M), *r2 = devz<float>(M);
  float *cosv = devz<float>((size_t)(N - 1) * HD / 2), *sinv = devz<float>((size_t)(N - 1) * HD / 2);
  uint16_t *xn1 = devz<uint16_t>((size_t)M * D), *att = devz<uint16_t>((size_t)M * D), *xmid = devz<uint16_t>((size_t)M * D),
           *xn2 = devz<uint16_t>((size_t)M * D), *gp = devz<uint16_t>((size_t)M * HID), *hh = devz<uint16_t>((size_t)M * HID),
           *out = devz<uint16_t>((size_t)M * D);
  if (!dx || !out) { fprintf(stderr, "hipMalloc failed\n"); return 1; }

  CK(vitpe_rope_axial_tables(dinv, cosv, sinv, G, HD / 2, st));
  CK(vitpe_pack_qkv_weights_wide(VITPE_BF16, dwqkv, dpack, D, HD, st));
  CK(vitpe_pack_weight_frags(VITPE_BF16, dwp, pwp, D, D, 192, 0, st));      // attn.proj.weight [192,192]
  CK(vitpe_pack_weight_frags(VITPE_BF16, dw1, pw1, HID, D, 192, 1, st));    // mlp.fc1.weight  [HID,192]
  CK(vitpe_pack_weight_frags(VITPE_BF16, dw2, pw2, D, HID, 32, 1, st));     // mlp.fc2.weight  [192,HID]
  CK(vitpe_layernorm_fwd(VITPE_BF16, dx, dg1, db1, NULL, m1, r1, M, D, 1e-5f, st));            // statistics only
  CK(vitpe_fused_attention_fwd_wide(VITPE_BF16, dx, dg1, db1, m1, r1, xn1, dpack, att, B, N, D, HD, VITPE_PE_ROPE_AXIAL, cosv, sinv,
                                    NULL, NULL, G, 0, 0, st));
  CK(vitpe_block_tail2_fwd(VITPE_BF16, att, dx, pwp, dbp, dg2, db2, xmid, m2, r2, xn2, pw1, dbf1, pw2, dbf2, gp, hh, out, NULL, NULL,
                           1e-5f, 1e-5f, M, D, HID, st));
  CK(hipStreamSynchronize(st));
  std::vector<uint16_t> ho((size_t)M * D);
  CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = 0; i < ho.size(); ++i) { const double v = bf2f(ho[i]); s1 += v; s2 += v * v * (double)((i % 7) + 1); }
  printf("abi_smoke ok sum=%.6f wsq=%.6f first=%.6f last=%.6f\n", s1, s2, bf2f(ho[0]), bf2f(ho.back()));
  // error behaviour of the boundary: an unsupported geometry is reported, never thrown
  const int e = vitpe_fused_attention_fwd(VITPE_BF16, xn1, dpack, att, 1, 17, 64, 32, VITPE_PE_NONE, NULL, NULL, NULL, NULL, 4, 0, 0, st);
  printf("unsupported geometry -> %d (hipErrorNotSupported = %d)\n", e, (int)hipErrorNotSupported);
  return e == (int)hipErrorNotSupported ? 0 : 2;
}
