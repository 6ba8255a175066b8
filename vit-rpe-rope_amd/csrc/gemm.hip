// MFMA GEMMs for the projection / MLP / patch-embed legs of the hot path.
//
//  gemm_nt : C[M,N] = epi( A[M,K] * W[N,K]^T )        forward linears and (with a transposed
//            weight shadow) the data-gradient GEMMs.   Replaces nn.Linear / nn.Conv2d-as-GEMM
//            (reference models/vit.py:35,37,91,164,248 and timm Mlp fc1/fc2, vit.py:118,124).
//  gemm_tn : dW[N,K] += dY[M,N]^T * X[M,K]             weight gradients (autograd of the same
//            linears), contraction over the token dimension, split over M across workgroups.
//
// Both are written once over T in {bf16, float} (common.h): bf16 = 16x16x32 MFMA throughput
// mode, float = exact fp32 16x16x4 MFMA for the 1e-4 parity gate.
#include "common.h"
#include "gemm_nt.h"
#include <stdlib.h>

namespace vitpe {

// Tile: 128 activation rows x 64 weight rows per workgroup, 128 bytes of K per stage,
// double-buffered in LDS, register-staged global loads (issue early / write late).
// MFMA orientation is swapped (A-operand = weight rows, B-operand = activation rows) so each
// lane ends up with 4 consecutive output columns of one row.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTArgs a) {
  constexpr int BM = 128, BN = 64, ROWB = 144;
  constexpr int BK = 128 / (int)sizeof(T);  // elements of K per stage (64 bf16 / 32 fp32)
  constexpr int CPS = BK / 32;              // K32 chunks per stage
  constexpr int CHN = CH<T>::n;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int EP_LD = BN + 4;             // fp32 epilogue staging row stride
  static_assert(2 * STAGE >= BM * EP_LD * 4, "epilogue staging must fit in the stage buffers");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int M = a.M, N = a.N, K = a.K;
  // 1-D grid, XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of tiles, with the N-tiles of one row panel adjacent -> the A panel is
  // fetched into ONE L2 once and the (small) weight matrix stays resident in every L2.
  const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM, total = ntn * ntm;
  int t;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int q = total >> 3, r = total & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
  const T* __restrict__ A = reinterpret_cast<const T*>(a.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.W);

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  Chunk16 ra[4], rw[2];
  const Chunk16 zero = {0u, 0u, 0u, 0u};
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
      const int gm = m0 + row, kk = k0 + cc * CHN;
      ra[i] = (gm < M && kk < K) ? *reinterpret_cast<const Chunk16*>(A + (size_t)gm * K + kk) : zero;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
      const int gn = n0 + row, kk = k0 + cc * CHN;
      rw[i] = (gn < N && kk < K) ? *reinterpret_cast<const Chunk16*>(W + (size_t)gn * K + kk) : zero;
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sW = sA + BM * ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
      *reinterpret_cast<Chunk16*>(sA + row * ROWB + cc * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i, row = q >> 3, cc = q & 7;
      *reinterpret_cast<Chunk16*>(sW + row * ROWB + cc * 16) = rw[i];
    }
  };

  const int nk = (K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const unsigned char* sA = smem + (kt & 1) * STAGE;
    const unsigned char* sW = sA + BM * ROWB;
#pragma unroll
    for (int cs = 0; cs < CPS; ++cs) {
      Frag<T> fw[4], fa[2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        fw[nt] = ld_frag(reinterpret_cast<const T*>(sW + (16 * nt + c) * ROWB) + cs * 32 + 8 * g);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        fa[mt] = ld_frag(reinterpret_cast<const T*>(sA + (32 * wave + 16 * mt + c) * ROWB) + cs * 32 + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma(fw[nt], fa[mt], acc[nt][mt]);
    }
    if (kt + 1 < nk) sstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> fp32 LDS tile [m][n] -> coalesced 8-column pieces -------
  float* ep = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = 32 * wave + 16 * mt + c, n = 16 * nt + 4 * g;
      *reinterpret_cast<f32x4*>(ep + m * EP_LD + n) = acc[nt][mt];
    }
  __syncthreads();
  T* __restrict__ C = reinterpret_cast<T*>(a.C);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i, m = q >> 3, n8 = (q & 7) * 8;
    const int gm = m0 + m, gn = n0 + n8;
    if (gm >= M || gn >= N) continue;  // N % 8 == 0 is required by the host wrapper
    float v[8];
    {
      f32x4 x = *reinterpret_cast<const f32x4*>(ep + m * EP_LD + n8);
      f32x4 y = *reinterpret_cast<const f32x4*>(ep + m * EP_LD + n8 + 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) { v[t] = x[t]; v[4 + t] = y[t]; }
    }
    if (EPI != EPI_GELU_BWD && a.bias != nullptr) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] += a.bias[gn + t];
    }
    size_t orow = (size_t)gm;
    if (EPI == EPI_PATCH) {
      const int b = gm / a.P, p = gm - b * a.P;
      orow = (size_t)b * a.Ntok + 1 + p;
      if (a.ape != nullptr) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += a.ape[(size_t)p * N + gn + t];
      }
      if (p == 0) {  // class-token row of this image (reference vit.py:253-254; APE skips it)
        float cv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) cv[t] = a.cls[gn + t];
        T* dst = C + (size_t)b * a.Ntok * N + gn;
#pragma unroll
        for (int h = 0; h < 8 / CHN; ++h)
          *reinterpret_cast<Chunk16*>(dst + h * CHN) = f32_to_chunk<T>(cv + h * CHN);
      }
    }
    const size_t off = orow * N + gn;
    if (EPI == EPI_BIAS_RESID) {
      const T* R = reinterpret_cast<const T*>(a.R);
      float rv[8];
#pragma unroll
      for (int h = 0; h < 8 / CHN; ++h)
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(R + off + h * CHN), rv + h * CHN);
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] += rv[t];
    }
    if (EPI == EPI_BIAS_GELU) {
      T* U = reinterpret_cast<T*>(a.U);
#pragma unroll
      for (int h = 0; h < 8 / CHN; ++h)
        *reinterpret_cast<Chunk16*>(U + off + h * CHN) = f32_to_chunk<T>(v + h * CHN);
      gelu_erf_x8(v);
    }
    if (EPI == EPI_GELU_BWD) {
      const T* U = reinterpret_cast<const T*>(a.U);
      float uv[8];
#pragma unroll
      for (int h = 0; h < 8 / CHN; ++h)
        chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(U + off + h * CHN), uv + h * CHN);
      gelu_erf_grad_mul_x8(v, uv);
    }
#pragma unroll
    for (int h = 0; h < 8 / CHN; ++h)
      *reinterpret_cast<Chunk16*>(C + off + h * CHN) = f32_to_chunk<T>(v + h * CHN);
  }
}

// ---------------------------------------------------------------------------------------
// Panel GEMM (second generation, used whenever N % 192 == 0): one workgroup = 12 waves computes a
// 144 x 192 output tile (3 x 4 waves, 3 x 3 MFMA tiles each), so for the N = 192 layers a tile is
// a set of FULL output rows: the activation panel is read from HBM exactly once and row-wise
// epilogues (LayerNorm statistics of the output) need no second kernel.  K is streamed in 128-byte
// slabs through two LDS buffers whose rows are XOR-swizzled (16-B slot ^= (row>>1)&7), which makes
// the ds_read_b128 fragment pattern bank-conflict free without padding; global loads run two
// slabs ahead in registers.  The epilogue works straight from the accumulators (each lane owns 4
// consecutive columns of a row).  Row panels are sized by the host so that the grid covers the
// 256 CUs evenly (M = 33280 -> 256 panels of 130 rows = 2 images).
struct GemmPanelArgs {
  const void* A;      // [M,K] T
  const void* W;      // [N,K] T
  void* C;            // [M,N] T
  const float* bias;  // [N] or null
  const void* R;      // residual [M,N] T (EPI_BIAS_RESID)
  void* U;            // pre-activation [M,N] T (EPI_BIAS_GELU out / EPI_GELU_BWD in)
  float* mean_out;    // optional (N == 192 only): LayerNorm statistics of the OUTPUT rows
  float* rstd_out;
  int M, N, K;
  int panel_rows;     // rows per workgroup panel (<= 144)
  float eps;
  // LNA (template): A is normalised on the way into LDS with these row statistics / parameters;
  // xn_out (nullable) receives the normalised A.   EPI_LN_BWD: the tile is dxn = dLN-output; the epilogue
  // turns it into dx = R + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dxn*gamma, xhat = (X-mean)*rstd
  // (X = ln_x, the LayerNorm INPUT rows) and accumulates dgamma / dbeta.
  const float* ln_gamma;
  const float* ln_beta;
  const float* ln_mean;
  const float* ln_rstd;
  void* xn_out;
  const void* ln_x;
  float* dgamma;
  float* dbeta;
};

// Two shapes of the same kernel (WR x WC waves, wave tile 48 rows x 192/WC columns):
//   3 x 4 = 12 waves, tile 144 x 192, 86 KB of LDS: one workgroup per CU;
//   2 x 3 =  6 waves, tile  96 x 192, 74 KB of LDS: an experiment (two workgroups per CU would let one's
//   epilogue overlap the other's MFMA phase) that does not pay -- see launch_gemm_panel_shape below.
template <typename T, int EPI, bool STATS, bool LNA, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) __attribute__((amdgpu_waves_per_eu(3)))
void gemm_panel_kernel(GemmPanelArgs a) {
  constexpr bool ROWMAP = STATS || (EPI == EPI_LN_BWD);   // 32 lanes per output row in the epilogue
  constexpr int NTHR = 64 * WR * WC;
  constexpr int BM = 48 * WR, BN = 192, ROWB = 128;
  constexpr int NTW = BN / 16 / WC;                       // 16-column tiles per wave: 3 (WC = 4) or 4 (WC = 3)
  constexpr int BK = ROWB / (int)sizeof(T), CPS = BK / 32, CHN = CH<T>::n;
  constexpr int STAGE = (BM + BN) * ROWB;                 // 43008 B (144 rows) / 36864 B (96 rows)
  constexpr int NCH = (BM + BN) * 8;                      // 16-B chunks per stage
  constexpr int NIT = (NCH + NTHR - 1) / NTHR;            // 4 / 6
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;
  const int M = a.M, N = a.N, K = a.K;
  const int ntn = N / BN;  // column tiles, all done by this workgroup (persistent over the row panel)
  const int panel = blockIdx.x;
  const int m0 = panel * a.panel_rows, m_end = min(M, m0 + a.panel_rows);
  const T* __restrict__ A = reinterpret_cast<const T*>(a.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(a.W);
  const Chunk16 zero = {0u, 0u, 0u, 0u};

  f32x4 acc[NTW][3];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  const int S = nk * ntn;  // flattened (column tile, K slab) stages
  auto gload = [&](Chunk16* r, int st) {
    const int n0 = (st / nk) * BN, k0 = (st % nk) * BK;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = tid + NTHR * i, row = q >> 3, cc = q & 7;
      const int kk = k0 + cc * CHN;
      Chunk16 v = zero;
      if (q < NCH && kk < K) {
        if (row < BM) {
          if (m0 + row < m_end) {
            v = *reinterpret_cast<const Chunk16*>(A + (size_t)(m0 + row) * K + kk);
            if (LNA) {
              const float mean = a.ln_mean[m0 + row], rstd = a.ln_rstd[m0 + row];
              float f[CHN];
              chunk_to_f32<T>(v, f);
#pragma unroll
              for (int h4 = 0; h4 < CHN / 4; ++h4) {
                const f32x4 gq = *reinterpret_cast<const f32x4*>(a.ln_gamma + kk + 4 * h4);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(a.ln_beta + kk + 4 * h4);
#pragma unroll
                for (int t = 0; t < 4; ++t) f[4 * h4 + t] = (f[4 * h4 + t] - mean) * rstd * gq[t] + bq[t];
              }
              v = f32_to_chunk<T>(f);
              if (a.xn_out != nullptr && st < nk)   // first column tile only
                __builtin_nontemporal_store(v, reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.xn_out) + (size_t)(m0 + row) * K + kk));
            }
          }
        } else {
          v = *reinterpret_cast<const Chunk16*>(W + (size_t)(n0 + row - BM) * K + kk);
        }
      }
      r[i] = v;
    }
  };
  auto sstore = [&](const Chunk16* r, int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = tid + NTHR * i, row = q >> 3, cc = q & 7;
      if (q < NCH) *reinterpret_cast<Chunk16*>(base + row * ROWB + ((cc ^ ((row >> 1) & 7)) << 4)) = r[i];
    }
  };
  auto compute = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sW = sA + BM * ROWB;
#pragma unroll
    for (int cs = 0; cs < CPS; ++cs) {
      Frag<T> fw[NTW], fa[3];
      // element offset 32cs + 8g  ->  16-B slot(s): bf16 slot = 4cs + g ; fp32 slots = 2g, 2g+1 (cs = 0)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int row = wn * (16 * NTW) + 16 * nt + c;
        const unsigned char* rp = sW + row * ROWB;
        const int sw = (row >> 1) & 7;
        if (sizeof(T) == 2) {
          fw[nt] = ld_frag(reinterpret_cast<const T*>(rp + (((4 * cs + g) ^ sw) << 4)));
        } else {
          fw[nt] = ld_frag2(reinterpret_cast<const T*>(rp + (((2 * g) ^ sw) << 4)),
                            reinterpret_cast<const T*>(rp + (((2 * g + 1) ^ sw) << 4)));
        }
      }
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const int row = wm * 48 + 16 * mt + c;
        const unsigned char* rp = sA + row * ROWB;
        const int sw = (row >> 1) & 7;
        if (sizeof(T) == 2) {
          fa[mt] = ld_frag(reinterpret_cast<const T*>(rp + (((4 * cs + g) ^ sw) << 4)));
        } else {
          fa[mt] = ld_frag2(reinterpret_cast<const T*>(rp + (((2 * g) ^ sw) << 4)),
                            reinterpret_cast<const T*>(rp + (((2 * g + 1) ^ sw) << 4)));
        }
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) mma(fw[nt], fa[mt], acc[nt][mt]);
    }
  };

  T* __restrict__ C = reinterpret_cast<T*>(a.C);
  // ---- epilogue: the C-layout (each lane 4 columns of 16 different rows) stores at < 1/2 of the
  // coalesced rate (measured), so the tile goes through LDS: in three passes of 48 rows the owning
  // waves park their fp32 accumulators in the (now idle) stage buffers, then all 768 threads apply
  // bias / residual / GELU on 8-column pieces with 16-byte coalesced loads and stores.
  constexpr int EP_LD = BN + 4;                      // fp32 row stride of the parked tile
  constexpr int RP = 16 * WR;                        // rows parked per pass
  constexpr int RPI = NTHR / 32;                     // ROWMAP: rows per iteration
  constexpr int EI = ROWMAP ? (RP + RPI - 1) / RPI : (RP * 24 + NTHR - 1) / NTHR;
  static_assert(RP * EP_LD * 4 <= 2 * STAGE, "parked tile must fit");
  float* ep = reinterpret_cast<float*>(smem);
  auto epilogue = [&](int n0) {
    const float invN = 1.0f / (float)BN;
    // EPI_LN_BWD (N == BN: this is the workgroup's only tile): column sums of this thread's 8 columns over
    // its rows; they live only inside the epilogue so the K loop carries no extra registers
    float lnacc_g[8], lnacc_b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { lnacc_g[t] = 0.f; lnacc_b[t] = 0.f; }
    // coalesced phase: 48 rows x 24 pieces of 8 columns.  Dense mapping (piece q = tid + 768 i) by
    // default; with STATS every row gets 32 lanes (8 idle) so its statistics reduce with shuffles.
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {           // pass = the mt tile every wave parks (static index: a
                                                     // runtime `pass` sends the accumulators to scratch, measured 2x)
      __syncthreads();                               // stage buffers / previous pass fully consumed
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        *reinterpret_cast<f32x4*>(ep + (16 * wm + c) * EP_LD + wn * (16 * NTW) + 16 * nt + 4 * g) = acc[nt][pass];
      __syncthreads();
#pragma unroll ((EPI == EPI_LN_BWD || (STATS && EI > 2)) ? 1 : EI)  // rolled where the iterations' temporaries would overlap and spill
      for (int i = 0; i < EI; ++i) {
        const int qd = tid + NTHR * i;
        const int row = ROWMAP ? (tid >> 5) + RPI * i : qd / 24;   // parked row: wave-row row/16, c = row%16
        const int pc = ROWMAP ? (tid & 31) : qd % 24;
        const bool live = ROWMAP ? (pc < 24 && row < RP) : (qd < RP * 24);
        const int gm = m0 + (row >> 4) * 48 + 16 * pass + (row & 15), gn = n0 + pc * 8;
        const bool ok = live && (gm < m_end);
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = 0.f;
        if (live) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8);
          const f32x4 y = *reinterpret_cast<const f32x4*>(ep + row * EP_LD + pc * 8 + 4);
#pragma unroll
          for (int t = 0; t < 4; ++t) { v[t] = x[t]; v[4 + t] = y[t]; }
        }
        if (EPI == EPI_LN_BWD) {
          // v = dxn piece.  Row reductions over the 32 lanes of the row, then dx and the dgamma/dbeta sums.
          const size_t off = (size_t)min(gm, m_end - 1) * N + gn;
          float xh[8], gv[8], gmv[8];
          float s1 = 0.f, s2 = 0.f;
          const float mean = a.ln_mean[min(gm, m_end - 1)], rstd = a.ln_rstd[min(gm, m_end - 1)];
          if (live) {
            float xv[8];
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.ln_x) + off + h * CHN), xv + h * CHN);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_gamma + gn);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.ln_gamma + gn + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { gmv[t] = g0[t]; gmv[4 + t] = g1[t]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              // the gradient as the unfused path sees it: dxn rounded to T
              v[t] = to_f32(from_f32<T>(v[t]));
              xh[t] = (xv[t] - mean) * rstd;
              gv[t] = v[t] * gmv[t];
              s1 += gv[t];
              s2 += gv[t] * xh[t];
            }
          }
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
          s1 *= invN;
          s2 *= invN;
          if (ok) {
            float rv[8], o8[8];
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.R) + off + h * CHN), rv + h * CHN);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              o8[t] = rstd * (gv[t] - s1 - xh[t] * s2) + rv[t];
              lnacc_g[t] += v[t] * xh[t];
              lnacc_b[t] += v[t];
            }
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              *reinterpret_cast<Chunk16*>(C + off + h * CHN) = f32_to_chunk<T>(o8 + h * CHN);
          }
        } else if (ok) {
          if (EPI != EPI_GELU_BWD && a.bias != nullptr) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + gn);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + gn + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] += b0[t]; v[4 + t] += b1[t]; }
          }
          const size_t off = (size_t)gm * N + gn;
          if (EPI == EPI_BIAS_RESID) {
            float rv[8];
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.R) + off + h * CHN), rv + h * CHN);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] += rv[t];
          }
          if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              __builtin_nontemporal_store(f32_to_chunk<T>(v + h * CHN), reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.U) + off + h * CHN));
            gelu_erf_x8(v);
          }
          if (EPI == EPI_GELU_BWD) {
            float uv[8];
#pragma unroll
            for (int h = 0; h < 8 / CHN; ++h)
              chunk_to_f32<T>(*reinterpret_cast<const Chunk16*>(reinterpret_cast<const T*>(a.U) + off + h * CHN), uv + h * CHN);
            gelu_erf_grad_mul_x8(v, uv);
          }
#pragma unroll
          for (int h = 0; h < 8 / CHN; ++h)
            *reinterpret_cast<Chunk16*>(C + off + h * CHN) = f32_to_chunk<T>(v + h * CHN);
        }
        // LayerNorm statistics of the output rows (values as stored, i.e. rounded to T): the 32 lanes
        // of a row reduce with shuffles (two-pass variance)
        if (STATS) {                                 // N == BN enforced by the host
          float sres = 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t) { v[t] = live ? to_f32(from_f32<T>(v[t])) : 0.f; sres += v[t]; }
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) sres += __shfl_xor(sres, o, 64);
          const float mean = sres * invN;
          float sq = 0.f;
          if (live) {
#pragma unroll
            for (int t = 0; t < 8; ++t) { const float d = v[t] - mean; sq += d * d; }
          }
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
          if (pc == 0 && gm < m_end) {
            a.mean_out[gm] = mean;
            a.rstd_out[gm] = 1.0f / sqrtf(sq * invN + a.eps);
          }
        }
      }
    }
    __syncthreads();                                 // parked tile consumed before the next slab is stored
    if (EPI == EPI_LN_BWD) {
      // thread (row-group tid>>5, piece pc = tid&31) holds sums for columns 8pc..8pc+7: reduce the 24 row
      // groups through LDS (the stage buffers are idle), then one atomic per column and workgroup
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);   // [RPI][2][192]
      const int pc = tid & 31, grp = tid >> 5;
      if (pc < 24) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          red[(grp * 2 + 0) * BN + pc * 8 + t] = lnacc_g[t];
          red[(grp * 2 + 1) * BN + pc * 8 + t] = lnacc_b[t];
        }
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        float sres = 0.f;
#pragma unroll
        for (int k = 0; k < RPI; ++k) sres += red[(k * 2 + which) * BN + col];
        atomicAdd((which == 0 ? a.dgamma : a.dbeta) + col, sres);
      }
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  // flattened stage loop, register prefetch ACROSS column-tile boundaries: the first slab(s) of the next
  // tile are in flight while the current tile's epilogue stores drain.  One workgroup per CU: two stages
  // ahead (two register sets); two workgroups per CU: one stage ahead (the co-resident workgroup covers
  // the rest, and the second register set would not fit the 168-VGPR budget of 3 waves per SIMD).
  if (WR * WC > 6) {
    Chunk16 ra[NIT], rb[NIT];
    gload(ra, 0);
    if (S > 1) gload(rb, 1);
    for (int st = 0; st < S; st += 2) {
      sstore(ra, 0);
      if (st + 2 < S) gload(ra, st + 2);
      __syncthreads();
      compute(0);
      if (st % nk == nk - 1) epilogue((st / nk) * BN);
      if (st + 1 < S) {
        sstore(rb, 1);
        if (st + 3 < S) gload(rb, st + 3);
        __syncthreads();
        compute(1);
        if ((st + 1) % nk == nk - 1) epilogue(((st + 1) / nk) * BN);
      }
    }
  } else {
    Chunk16 ra[NIT];
    gload(ra, 0);
    for (int st = 0; st < S; ++st) {
      sstore(ra, st & 1);
      if (st + 1 < S) gload(ra, st + 1);
      __syncthreads();
      compute(st & 1);
      if (st % nk == nk - 1) epilogue((st / nk) * BN);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Weight gradient: dW[n][k] += sum_m dY[m][n] * X[m][k]   (+ dbias[n] += sum_m dY[m][n])
// Tile TN=128 (n) x TK (k) per workgroup, 2x2 waves, RPS token rows per stage staged
// row-major (as they lie in HBM, coalesced) and consumed through transposed LDS reads.
// grid = (ceil(N/TN), ceil(K/TK), splits); each z-slice reduces its own token range and
// adds its partial tile with fp32 atomics (summation order across slices is not fixed).
struct GemmTNArgs {
  const void* dY;  // [M,N] T
  const void* X;   // [M,K] T
  float* dW;       // [N,K] fp32, accumulated into
  float* dbias;    // [N] fp32 or null, accumulated into
  int M, N, K;
  int rows_per_split;
  long long sn, sk;   // output strides: dW element (n,k) lives at dW[n*sn + k*sk] (lets the roles be swapped)
  int bias_from_x;    // dbias = colsum(X) (length K) instead of colsum(dY) (length N)
};

template <typename T, int TK>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs a) {
  constexpr int TN = 128;
  constexpr int RPS = 128 / (int)sizeof(T);  // token rows per stage: 64 bf16 / 32 fp32
  constexpr int CPS = RPS / 32;
  constexpr int CHN = CH<T>::n;
  constexpr int LDY = TN + Pad<T>::elems, LDX = TK + Pad<T>::elems;
  constexpr int YCH = TN / CHN, XCH = TK / CHN;          // 16-B chunks per row
  constexpr int YIT = RPS * YCH / 256, XIT = RPS * XCH / 256;
  static_assert(RPS * YCH % 256 == 0 && RPS * XCH % 256 == 0, "stage must divide over 256 threads");
  constexpr int KT = TK / 32;  // k tiles (16 wide) per wave (each wave owns half of TK)
  __shared__ __attribute__((aligned(16))) T sY[RPS * LDY];
  __shared__ __attribute__((aligned(16))) T sX[RPS * LDX];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int M = a.M, N = a.N, K = a.K;
  // 1-D grid, XCD-local order: every (n,k) tile of one token slice lands on the same XCD back to back, so
  // the slice's dY / X rows are fetched into that L2 once and re-read from there by the other tiles
  const int ntn = (N + TN - 1) / TN, ntk = (K + TK - 1) / TK, ntile = ntn * ntk;
  int t;
  {
    const int total = (int)gridDim.x, id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int q = total >> 3, r = total & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int zslice = t / ntile, tile = t % ntile;
  const int n0 = (tile % ntn) * TN, k0 = (tile / ntn) * TK;
  const int mbeg = zslice * a.rows_per_split;
  const int mend = min(M, mbeg + a.rows_per_split);
  const T* __restrict__ dY = reinterpret_cast<const T*>(a.dY);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.X);

  f32x4 acc[4][KT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = (a.dbias != nullptr) && (a.bias_from_x ? n0 == 0 : k0 == 0);
  const Chunk16 zero = {0u, 0u, 0u, 0u};

  Chunk16 ry[YIT], rx[XIT];
  auto gload = [&](int mb) {
#pragma unroll
    for (int i = 0; i < YIT; ++i) {
      const int q = tid + 256 * i, row = q / YCH, cc = q % YCH;
      const int gm = mb + row, gn = n0 + cc * CHN;
      ry[i] = (gm < mend && gn < N) ? *reinterpret_cast<const Chunk16*>(dY + (size_t)gm * N + gn) : zero;
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int q = tid + 256 * i, row = q / XCH, cc = q % XCH;
      const int gm = mb + row, gk = k0 + cc * CHN;
      rx[i] = (gm < mend && gk < K) ? *reinterpret_cast<const Chunk16*>(X + (size_t)gm * K + gk) : zero;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < YIT; ++i) {
      const int q = tid + 256 * i, row = q / YCH, cc = q % YCH;
      *reinterpret_cast<Chunk16*>(sY + row * LDY + cc * CHN) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int q = tid + 256 * i, row = q / XCH, cc = q % XCH;
      *reinterpret_cast<Chunk16*>(sX + row * LDX + cc * CHN) = rx[i];
    }
  };

  if (mbeg < mend) gload(mbeg);
  for (int mb = mbeg; mb < mend; mb += RPS) {
    __syncthreads();  // previous stage fully consumed
    sstore();
    __syncthreads();
    if (mb + RPS < mend) gload(mb + RPS);  // next stage in flight under the MFMAs
#pragma unroll
    for (int cs = 0; cs < CPS; ++cs) {
      const int rb0 = cs * 32 + 8 * g, rb1 = rb0 + 4;
      Frag<T> fy[4], fx[KT];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fy[nt] = ld_frag_tr(sY, LDY, rb0, rb1, wn * 64 + 16 * nt);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) fx[kt] = ld_frag_tr(sX, LDX, rb0, rb1, wk * (TK / 2) + 16 * kt);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) mma(fy[nt], fx[kt], acc[nt][kt]);
    }
    if (do_bias && tid < (a.bias_from_x ? TK : TN)) {
      float s = 0.f;
      if (a.bias_from_x) { for (int r = 0; r < RPS; ++r) s += to_f32(sX[r * LDX + tid]); }
      else { for (int r = 0; r < RPS; ++r) s += to_f32(sY[r * LDY + tid]); }
      bsum += s;
    }
  }

#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gn = n0 + wn * 64 + 16 * nt + 4 * g + r;
        const int gk = k0 + wk * (TK / 2) + 16 * kt + c;
        if (gn < N && gk < K) atomicAdd(a.dW + gn * a.sn + gk * a.sk, acc[nt][kt][r]);
      }
  if (do_bias) {
    if (a.bias_from_x) { if (tid < TK && k0 + tid < K) atomicAdd(a.dbias + k0 + tid, bsum); }
    else if (tid < TN && n0 + tid < N) atomicAdd(a.dbias + n0 + tid, bsum);
  }
}

}  // namespace vitpe

using namespace vitpe;

template <typename T>
static int launch_gemm_nt(int epi, const GemmNTArgs& a, hipStream_t s) {
  dim3 grid(((a.M + 127) / 128) * ((a.N + 63) / 64)), block(256);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS>), grid, block, 0, s, a); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS_GELU>), grid, block, 0, s, a); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS_RESID>), grid, block, 0, s, a); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_PATCH>), grid, block, 0, s, a); break;
    case EPI_GELU_BWD: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GELU_BWD>), grid, block, 0, s, a); break;
    default: return (int)hipErrorInvalidValue;
  }
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_gemm_nt(int dtype, int epi, const void* A, const void* W, void* C,
                             const float* bias, const void* R, void* U, const float* ape,
                             const float* cls, int M, int N, int K, int P, int Ntok,
                             hipStream_t stream) {
  VITPE_REQUIRE(A && W && C && M >= 0 && N > 0 && K > 0);
  VITPE_REQUIRE(N % 8 == 0);
  VITPE_REQUIRE(K % (dtype == 1 ? 8 : 4) == 0);
  VITPE_REQUIRE(dtype == 0 || dtype == 1);
  if (epi == EPI_BIAS_RESID) VITPE_REQUIRE(R != nullptr);
  if (epi == EPI_BIAS_GELU || epi == EPI_GELU_BWD) VITPE_REQUIRE(U != nullptr);
  if (epi == EPI_PATCH) VITPE_REQUIRE(cls != nullptr && P > 0 && Ntok == P + 1 && M % P == 0);
  if (M == 0) return 0;
  GemmNTArgs a{A, W, C, bias, R, U, ape, cls, M, N, K, P, Ntok};
  if (gemm2d_takes(dtype, epi, M, N, K)) return gemm2d_launch(epi, a, stream);
  return dtype == 1 ? launch_gemm_nt<bf16>(epi, a, stream) : launch_gemm_nt<float>(epi, a, stream);
}

// panel rows: as many panels as a multiple of the workgroup slots (CUs x resident workgroups per CU), each at
// most `bm` rows
static int panel_rows_for(int M, int bm, int slots) {
  const int waves = (M + slots * bm - 1) / (slots * bm);
  const int npanels = slots * waves;
  int rows = (M + npanels - 1) / npanels;
  return rows < 16 ? 16 : rows;
}

// 3 x 4 waves, 144-row tiles, one workgroup per CU.  (A 2 x 3-wave, 96-row shape meant to run two workgroups per CU was
// measured in round 1: its 74 KB of LDS is above the 64 KB up to which two workgroups are co-resident, so it ran one
// 6-wave workgroup per CU and was 1.5x slower -- fc1+GELU 58 vs 38 us; no longer instantiated.)
template <typename T, int WR, int WC>
static int launch_gemm_panel_shape(int epi, GemmPanelArgs a, hipStream_t s) {
  constexpr int WGS_PER_CU = (WR * WC <= 6) ? 2 : 1;
  a.panel_rows = panel_rows_for(a.M, 48 * WR, 256 * WGS_PER_CU);
  const int npanels = (a.M + a.panel_rows - 1) / a.panel_rows;
  dim3 grid(npanels), block(64 * WR * WC);
#define VITPE_PANEL(EPI, ST, LN) hipLaunchKernelGGL((gemm_panel_kernel<T, EPI, ST, LN, WR, WC>), grid, block, 0, s, a)
  if (a.mean_out != nullptr) {
    switch (epi) {
      case EPI_BIAS: VITPE_PANEL(EPI_BIAS, true, false); break;
      case EPI_BIAS_RESID: VITPE_PANEL(EPI_BIAS_RESID, true, false); break;
      default: return (int)hipErrorInvalidValue;
    }
    VITPE_CHECK_LAUNCH();
  }
  if (a.ln_gamma != nullptr && epi != EPI_LN_BWD) {   // LayerNorm fused into the A-operand staging
    switch (epi) {
      case EPI_BIAS: VITPE_PANEL(EPI_BIAS, false, true); break;
      case EPI_BIAS_GELU: VITPE_PANEL(EPI_BIAS_GELU, false, true); break;
      default: return (int)hipErrorInvalidValue;
    }
    VITPE_CHECK_LAUNCH();
  }
  switch (epi) {
    case EPI_BIAS: VITPE_PANEL(EPI_BIAS, false, false); break;
    case EPI_BIAS_GELU: VITPE_PANEL(EPI_BIAS_GELU, false, false); break;
    case EPI_BIAS_RESID: VITPE_PANEL(EPI_BIAS_RESID, false, false); break;
    case EPI_GELU_BWD: VITPE_PANEL(EPI_GELU_BWD, false, false); break;
    case EPI_LN_BWD: VITPE_PANEL(EPI_LN_BWD, false, false); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef VITPE_PANEL
  VITPE_CHECK_LAUNCH();
}

template <typename T>
static int launch_gemm_panel(int epi, GemmPanelArgs a, hipStream_t s) {
  return launch_gemm_panel_shape<T, 3, 4>(epi, a, s);
}

// C = epi(A W^T) like vitpe_gemm_nt (no EPI_PATCH), plus optional LayerNorm statistics of the output rows
// (mean_out/rstd_out non-null requires N == 192).  Falls back to the generic tile kernel when N % 192 != 0.
extern "C" int vitpe_linear(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                            const void* R, void* U, float* mean_out, float* rstd_out, float eps, int M, int N, int K,
                            hipStream_t stream) {
  VITPE_REQUIRE(A && W && C && M >= 0 && N > 0 && K > 0 && (dtype == 0 || dtype == 1));
  VITPE_REQUIRE(epi != EPI_PATCH && N % 8 == 0 && K % (dtype == 1 ? 8 : 4) == 0);
  VITPE_REQUIRE((mean_out == nullptr) == (rstd_out == nullptr));
  if (epi == EPI_BIAS_RESID) VITPE_REQUIRE(R != nullptr);
  if (epi == EPI_BIAS_GELU || epi == EPI_GELU_BWD) VITPE_REQUIRE(U != nullptr);
  if (mean_out) VITPE_REQUIRE(N == 192);
  if (M == 0) return 0;
  if (N % 192 != 0) {
    VITPE_REQUIRE(mean_out == nullptr);
    return vitpe_gemm_nt(dtype, epi, A, W, C, bias, R, U, nullptr, nullptr, M, N, K, 0, 0, stream);
  }
  // The panel kernel wants M >> 256 x 144 rows: with fewer rows its panels are mostly padding and every one of
  // them re-reads the whole weight (ViT-B/16 at batch 64: M = 12 608 -> 50-row panels, 240-300 TFLOP/s), while the
  // 2-D tiled kernel runs the same shapes at 380-510 TFLOP/s (tools/kbench_imnet.py).  Big weights, few rows:
  if (mean_out == nullptr && panel_rows_for(M, 144, 256) < 112 && N >= 384 && K >= 384)
    return vitpe_gemm_nt(dtype, epi, A, W, C, bias, R, U, nullptr, nullptr, M, N, K, 0, 0, stream);
  GemmPanelArgs a{};
  a.A = A; a.W = W; a.C = C; a.bias = bias; a.R = R; a.U = U; a.mean_out = mean_out; a.rstd_out = rstd_out;
  a.M = M; a.N = N; a.K = K; a.eps = eps;
  return dtype == 1 ? launch_gemm_panel<bf16>(epi, a, stream) : launch_gemm_panel<float>(epi, a, stream);
}

// fc1-style linear with the preceding LayerNorm fused into the A staging (vit.py:116,124):
// C = epi(LN(X) W^T), X raw [M,K], mean/rstd its row statistics; xn_out (nullable) receives LN(X).
// epi in {EPI_BIAS, EPI_BIAS_GELU}; N % 192 == 0.
extern "C" int vitpe_linear_ln(int dtype, int epi, const void* X, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, void* xn_out, const void* W, void* C,
                               const float* bias, void* U, int M, int N, int K, hipStream_t stream) {
  VITPE_REQUIRE(X && gamma && beta && mean && rstd && W && C && M >= 0 && (dtype == 0 || dtype == 1));
  VITPE_REQUIRE((epi == EPI_BIAS || epi == EPI_BIAS_GELU) && N % 192 == 0 && K % (dtype == 1 ? 8 : 4) == 0);
  if (epi == EPI_BIAS_GELU) VITPE_REQUIRE(U != nullptr);
  if (M == 0) return 0;
  GemmPanelArgs a{};
  a.A = X; a.W = W; a.C = C; a.bias = bias; a.U = U; a.M = M; a.N = N; a.K = K;
  a.ln_gamma = gamma; a.ln_beta = beta; a.ln_mean = mean; a.ln_rstd = rstd; a.xn_out = xn_out;
  return dtype == 1 ? launch_gemm_panel<bf16>(epi, a, stream) : launch_gemm_panel<float>(epi, a, stream);
}

// data-gradient GEMM with the LayerNorm backward fused into its epilogue (N == 192 == LayerNorm width):
// dx = dres + LN'(dY Wt^T) for the LayerNorm whose input rows are x (statistics mean/rstd, weight gamma);
// dgamma / dbeta accumulated (fp32 atomics, one per column and workgroup).
extern "C" int vitpe_linear_lnbwd(int dtype, const void* dY, const void* Wt, void* dx, const void* x, const float* mean,
                                  const float* rstd, const float* gamma, const void* dres, float* dgamma, float* dbeta,
                                  int M, int K, hipStream_t stream) {
  VITPE_REQUIRE(dY && Wt && dx && x && mean && rstd && gamma && dres && dgamma && dbeta && M >= 0);
  VITPE_REQUIRE((dtype == 0 || dtype == 1) && K % (dtype == 1 ? 8 : 4) == 0);
  if (M == 0) return 0;
  GemmPanelArgs a{};
  a.A = dY; a.W = Wt; a.C = dx; a.R = dres; a.M = M; a.N = 192; a.K = K;
  a.ln_gamma = gamma; a.ln_mean = mean; a.ln_rstd = rstd; a.ln_x = x; a.dgamma = dgamma; a.dbeta = dbeta;
  return dtype == 1 ? launch_gemm_panel<bf16>(EPI_LN_BWD, a, stream) : launch_gemm_panel<float>(EPI_LN_BWD, a, stream);
}

template <typename T>
static int launch_gemm_tn(GemmTNArgs a, int splits, hipStream_t s) {
  constexpr int RPS = 128 / (int)sizeof(T);
  int rps = (a.M + splits - 1) / splits;
  rps = (rps + RPS - 1) / RPS * RPS;
  a.rows_per_split = rps;
  const int nz = (a.M + rps - 1) / rps;
  dim3 block(256);
  // (128 x 192 tiles were measured 15 % slower: LDS-read bound on the transposed fragment reads)
  if (a.K % 128 == 0 || a.K > 192) {
    dim3 grid(((a.N + 127) / 128) * ((a.K + 127) / 128) * nz);
    hipLaunchKernelGGL((gemm_tn_kernel<T, 128>), grid, block, 0, s, a);
  } else {
    dim3 grid(((a.N + 127) / 128) * ((a.K + 63) / 64) * nz);
    hipLaunchKernelGGL((gemm_tn_kernel<T, 64>), grid, block, 0, s, a);
  }
  VITPE_CHECK_LAUNCH();
}

extern "C" int vitpe_gemm_tn(int dtype, const void* dY, const void* X, float* dW, float* dbias,
                             int M, int N, int K, int splits, hipStream_t stream) {
  VITPE_REQUIRE(dY && X && dW && M >= 0 && N > 0 && K > 0 && splits >= 1);
  VITPE_REQUIRE(dtype == 0 || dtype == 1);
  VITPE_REQUIRE(N % (dtype == 1 ? 8 : 4) == 0 && K % (dtype == 1 ? 8 : 4) == 0);
  if (M == 0) return 0;
  GemmTNArgs a{dY, X, dW, dbias, M, N, K, 0, (long long)K, 1LL, 0};
  return dtype == 1 ? launch_gemm_tn<bf16>(a, splits, stream) : launch_gemm_tn<float>(a, splits, stream);
}
