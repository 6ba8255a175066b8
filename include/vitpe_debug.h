/* vitpe_debug.h -- developer instrumentation exported by libvitpe.so next to the product ABI (include/vitpe.h).
 * Nothing here is on the hot path or part of the drop-in boundary: primitive self-tests of the MFMA operand maps and
 * the residency / phase-census launches used by tools/census*.py.  The census variants are separate template
 * instantiations; the product kernels carry no stamps.                                                             */
#ifndef VITPE_DEBUG_H
#define VITPE_DEBUG_H
#include "vitpe.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitive self-test (MFMA operand maps, transposed LDS read) ------------------------- */
int vitpe_selftest_mma(int dtype, const void* A, const void* Bt, const void* Brow, float* C_row,
                       float* C_tr, vitpe_stream_t stream);

/* debug: resident workgroups/CU the runtime computes for attention kernel `which` (0 fwd rope, 1 fwd plain, 2 bwd rope) */
int vitpe_debug_attn_occupancy(int which);
/* phase census of the grouped weight-gradient kernel (a separate template instantiation with s_memtime stamps) */
int vitpe_debug_wgrad_census(int dtype, const vitpe_wgrad_problem* problems, int nprob, unsigned long long* census,
                             vitpe_stream_t stream);
/* phase census of the fused attention forward (bf16, d=192, N=65, rope-axial; a separate instantiation with stamps):
 * census[(workgroup * 16 + wave) * 8 + slot] = s_memtime at 0 start, 1 tokens staged, 2 barrier passed, 3 v projected,
 * 4 k projected, 5 q projected, 6 end                                                                              */
int vitpe_debug_attn_census(const void* xn, const void* wqkv, void* out, const float* cos, const float* sin,
                            int B, unsigned long long* census, vitpe_stream_t stream);
/* the same for the wide forward (csrc/attn32.hip; wqkv_wide = vitpe_pack_qkv_weights_wide): 16 slots per wave: 0 start,
 * 1 staged, 2 barrier passed, 3 projection k-loop done, 4 operand fragments built, 5 patch queries done, 6 end, 8 cycles in
 * the k-loop's barriers, 9 / 10 s_memrealtime (100 MHz) at start / end, 11 cycles waiting for the own LDS-DMA pieces; exp: timing experiments with
 * WRONG results (1 no 65th-token work in the k-loop, 2 no k-loop barriers), 0 = the real kernel                                       */
int vitpe_debug_attn32_census(const void* xn, const void* wqkv_wide, void* out, const float* cos, const float* sin,
                              int B, unsigned long long* census, int exp, vitpe_stream_t stream);
/* phase census of the second-generation block tail (training instantiation with stamps; bf16, D = 192):
 * census[(workgroup * 9 + wave) * 16 + slot] = s_memtime at 0 start, 1 first slab landed, 2 proj product done,
 * 3 LayerNorm2 epilogue done, 4 period-0 barrier passed, 5 period 0 done, 6 periods 1.. done, 7 last barrier passed,
 * 8 last fc2 product done, 9 end; slots 10 / 11 / 12 = ticks summed over periods 1.. in the barrier wait, the fc1
 * product, the interleaved {fc2 || GELU} step.  exp: timing experiments with WRONG results (1 quarter of the LDS
 * fragment reads, 2 no erf, 4 no hidden-layer stores, 7 all), 0 = the real kernel.                                                                        */
int vitpe_debug_tail2_census(const void* attn_out, const void* x_in, const void* Wp_packed, const float* bp,
                             const float* gamma, const float* beta, void* x_mid, float* mean2, float* rstd2,
                             void* xn_out, const void* W1_packed, const float* b1, const void* W2_packed,
                             const float* b2, void* gp_out, void* h_out, void* out, float* mean_out, float* rstd_out,
                             int M, int HID, unsigned long long* census, int exp, vitpe_stream_t stream);
/* Tile height of the big-tile bf16 GEMM behind vitpe_gemm_nt / vitpe_linear (csrc/gemm2d.hip): 0 = the host's choice per
 * shape, 4 / 5 / 6 = (32 mt)-row tiles for every launch.  Tests and A/B measurements.                                */
int vitpe_debug_set_gemm2d_mt(int mt);
/* 0: problem lists that qualify for the 192 x 384-block weight-gradient kernel (bf16, plain X, N % 192 == 0, K % 384 == 0)
 * stay on the 192 x 192 kernel; 1 (default): they take the wide one.  Tests and A/B measurements (tools/kb_wgrad.py). */
int vitpe_debug_set_wgrad_wide(int on);
/* Read-only streaming kernel: `bytes` of src with 16-B loads, `depth` (1 / 4 / 8) independent loads in flight per lane,
 * `workgroups` x 256 threads, grid-stride; writes nothing.  The read ceiling of this box (tools/membw.py).        */
int vitpe_debug_read_bw(const void* src, long long bytes, int depth, int workgroups, unsigned* sink, vitpe_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif /* VITPE_DEBUG_H */
