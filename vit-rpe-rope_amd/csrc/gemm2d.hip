// Big-tile bf16 GEMM for the many-row, wide-weight linears of the ImageNet geometry (ViT-B/16: M = B x 197 tokens,
// N, K in {768, 2304, 3072}):   C[M,N] = epi( A[M,K] * W[N,K]^T ),  same argument block and epilogues as gemm_nt
// (gemm.hip), which hands these shapes over (gemm2d_takes).  Reference: nn.Linear qkv / proj (models/vit.py:35,37,91),
// timm Mlp fc1 / fc2 (vit.py:118,124) and, on transposed weight shadows, their data gradients.
//
// One workgroup = 8 waves (2 x 4) computes a (32 MT) x 256 output tile, MT in {4, 5, 6}: the wave tile is 16 MT
// activation rows x 64 weight rows, 4 MT MFMAs per 32-deep k step against MT + 4 fragment reads.  MT is chosen per
// shape by the host so that (tiles / 256 CUs rounded up) x MT -- the launch's critical path -- is smallest (M = 12 608:
// MT = 5 puts the N = 768 layers on 237 of the 256 CUs in ONE round; 256-row tiles would use 150).
//
// K moves in 64-element (128-byte) steps through LDS stage buffers (three for MT <= 5, else two) filled by LDS-DMA (global_load_lds, 16 B per lane,
// no staging registers, no ds_write pass): a wave instruction lands 8 rows x 128 B lane-linearly, so the bank swizzle
// (16-B slot ^= (row >> 1) & 7 -- conflict-free for the ds_read_b128 fragment pattern, as in gemm_panel_kernel) is applied
// to the SOURCE address and again on the fragment read.  One barrier per K step: wait for this wave's pieces of step kt,
// barrier (now everybody's pieces have landed and everybody has left step kt - 1's buffer), issue step kt + 1 (kt + 2 with
// three buffers: a K step is ~1.3 K cycles of MFMA work, less than the DMA's latency), compute kt.
//
// The MFMA orientation is swapped (A operand = weight rows, B operand = activation rows): a lane's accumulator holds 4
// consecutive output columns of one row.  Epilogue, in registers: one v_permlane16_swap per dword between lane groups g and
// g ^ 1 gives every lane 8 consecutive columns (fp32), bias / residual / GELU are applied on those and the row leaves as
// 16-byte pieces (64 contiguous bytes per 4 lanes).  The kernel is persistent (one workgroup per CU walks its tiles): the
// next tile's first stages are issued before the epilogue, and the epilogue's stores stay in flight under the next tile's
// first K steps (counted waits, see sync()).
#include "common.h"
#include "gemm_nt.h"
#include <stdlib.h>
#include <type_traits>

namespace vitpe {

#ifndef T2D_SUPER
#define T2D_SUPER 1   // supertile order for wide outputs (tile_of); 0: row-panel-major (A/B builds)
#endif

template <int MT, int EPI>
__global__ __launch_bounds__(512) void gemm2d_kernel(GemmNTArgs a) {
  typedef bf16 T;
  constexpr int BM = 32 * MT, BN = 256, BK = 64;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int AG = BM / 8, WG_ = BN / 8;            // 8-row groups (one wave instruction each) of the A / W stage
  constexpr int AI = (AG + 7) / 8, WI = WG_ / 8;      // instructions per wave and stage
  constexpr int NBUF = (3 * STAGE <= 160 * 1024) ? 3 : 2;
  constexpr int AHEAD = NBUF - 1;                     // stages in flight
  static_assert(AI + WI < 16, "stage wait counts fit 4 bits");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NBUF * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int M = a.M, N = a.N, K = a.K;
  const int ntn = N / BN, ntm = (M + BM - 1) / BM, total = ntn * ntm;
  // XCD-aware order: an XCD (workgroup index % 8; the grid is a multiple of 8 or the whole tile list) gets a contiguous run
  // of tiles, the column tiles of a row panel adjacent
  auto tile_of = [&](int id, int& m0, int& n0) {
    const int xcd = id & 7, slot = id >> 3;
    const int q = total >> 3, r = total & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    // Wide outputs (more than four column tiles): the tile order inside the run goes by SUPERTILES of 8 row panels x 4
    // column tiles -- what an XCD's 32 CUs run at one time then needs 8 A panels + 4 W tiles (3.5 MB at K = 768: its L2),
    // and the next column group finds the A panels still there.  Row-panel-major order put 2.7 panels x ALL column tiles in
    // flight: the whole weight (4.7 MB at N = 3072) went through every XCD's L2 once per wave of tiles (PMC: 335 MB
    // fetched + written per fc1 launch for 179 MB of operands).
    if (T2D_SUPER && (ntn & 3) == 0 && ntn > 4) {
      constexpr int GH = 8;
      const int per = GH * ntn, pg = t / per, rr0 = t - pg * per;
      const int gh = min(GH, ntm - GH * pg);
      const int cg = rr0 / (gh * 4), rr = rr0 - cg * gh * 4;
      m0 = (GH * pg + (rr >> 2)) * BM;
      n0 = (cg * 4 + (rr & 3)) * BN;
      return;
    }
    m0 = (t / ntn) * BM;
    n0 = (t % ntn) * BN;
  };

  // ---- LDS-DMA sources: lane l of a wave instruction fills row (l >> 3) of its 8-row group, physical slot l & 7
  const T* pa[AI];
  const T* pw[WI];
  auto sources = [&](int m0, int n0) {
    const int rl = lane >> 3, ps = lane & 7;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = (8 * i + wave) * 8 + rl;
      int gm = m0 + row;
      gm = gm < M ? gm : M - 1;                       // rows past M: any valid row (masked in the epilogue)
      pa[i] = reinterpret_cast<const T*>(a.A) + (size_t)gm * K + 8 * (ps ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = (8 * i + wave) * 8 + rl;
      pw[i] = reinterpret_cast<const T*>(a.W) + (size_t)(n0 + row) * K + 8 * (ps ^ ((row >> 1) & 7));
    }
  };
  auto stage = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sW = sA + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      if (8 * i + wave < AG)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + k0),
                                         (__attribute__((address_space(3))) void*)(sA + (8 * i + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[i] + k0),
                                       (__attribute__((address_space(3))) void*)(sW + (8 * i + wave) * 1024), 16, 0, 0);
  };

  // fragment addresses: row = 16 x (tile) + c  ->  (row >> 1) & 7 == (c >> 1) & 7 ; slot = 4 ks + g
  const int sw = (c >> 1) & 7;
  const int off0 = c * 128 + ((g ^ sw) << 4), off1 = c * 128 + (((4 + g) ^ sw) << 4);
  f32x4 acc[4][MT];
  // A K step is two 32-deep k steps; their fragment sets live in two register sets (P: k step 0, Q: k step 1) so that the
  // LDS reads of one run under the MFMAs of the other -- ACROSS the barrier too: the MFMAs of (kt - 1, k step 1) are issued
  // after step kt's barrier, next to the reads of (kt, k step 0).  (All eight waves leave a barrier in the same phase, and
  // the CU's LDS read time per K step, 8 waves x 2 x (MT + 4) KB at 128 B/clk, is about its MFMA time: read-then-multiply
  // ran the main loop at 40 % of the matrix pipe.)
  Frag<T> fwP[4], faP[MT], fwQ[4], faQ[MT];
  auto rd = [&](int buf, int ks, Frag<T>* fw, Frag<T>* fa) {
    const unsigned char* sA = smem + buf * STAGE + wm * (16 * MT) * 128;
    const unsigned char* sW = smem + buf * STAGE + A_BYTES + wn * 64 * 128;
    const int off = ks ? off1 : off0;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fw[nt] = ld_frag(reinterpret_cast<const T*>(sW + nt * 2048 + off));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = ld_frag(reinterpret_cast<const T*>(sA + mt * 2048 + off));
  };
  auto mm = [&](const Frag<T>* fw, const Frag<T>* fa) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) mma(fw[nt], fa[mt], acc[nt][mt]);
  };
  // an MFMA, then a fragment read, (MT + 4) times; the rest of the 4 MT MFMAs after.  (MFMA first: the compiler waits
  // lgkmcnt(0) before the first MFMA of a half step -- its operands were read half a step ago -- and with a read in front
  // of it that wait would be a stall for the NEW read.)
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < MT + 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - (MT + 4), 0);
  };

  const int nk = K / BK;
  const bool full = __builtin_amdgcn_readfirstlane(8 * (AI - 1) + wave) < AG;   // this wave issues AI (else AI - 1) A pieces per stage
  // wait for this wave's pieces of stage kt (leaving the younger stage's outstanding) and for its own fragment reads; then
  // the barrier: everybody's pieces of stage kt have landed and everybody has left the buffer stage kt + AHEAD goes to.
  // (A raw s_barrier: __syncthreads() would drain the DMA in flight.)  Step 0 waits for everything: the previous tile's
  // stores are in the same counter.
#define VITPE_WAITVM(n) __builtin_amdgcn_s_waitcnt(0x0070 | ((n) & 15) | (((n) >> 4) << 14))   /* vmcnt(n) lgkmcnt(0) */
  // vmcnt counts loads, LDS-DMA pieces and stores alike and retires them in issue order (MI355X_MICROARCH.md, `s_waitcnt
  // vmcnt(N)`: "all but the wave's N youngest vector-memory operations are done"; flat_* excepted, none here).  A tile's first stages are issued
  // BEFORE the previous tile's epilogue, so at steps 0 and 1 (AHEAD == 2; step 0 with two buffers) the wait leaves that
  // epilogue's EP_OPS loads / stores outstanding as well: the stores drain under the first K steps instead of stalling the
  // tile start.  ep_pending is EP_OPS, or 0 for the first tile and after a ragged tile (whose waves skip instructions for
  // rows past M and therefore end their epilogue with vmcnt(0)).
  // (EPI_BIAS: 2 MT stores; EPI_BIAS_GELU: 4 MT stores; the epilogues with an input operand: 2 MT stores + 2 (MT - 1) row
  //  loads, two fewer than the constant -- harmless: their last pass has consumed a load YOUNGER than the next tile's
  //  first stages, so those have retired whatever the count says)
  constexpr int EP_OPS = (EPI == EPI_BIAS ? 1 : 2) * 2 * MT;
  static_assert(EP_OPS + AI + WI < 64, "vmcnt is a 6-bit counter");
  int ep_pending = 0;
  auto sync = [&](int kt) {
    if (AHEAD == 2 && kt + 1 < nk) {
      if (kt < 2 && ep_pending != 0) {
        if (full) VITPE_WAITVM(EP_OPS + AI + WI);
        else VITPE_WAITVM(EP_OPS + AI - 1 + WI);
      } else {
        if (full) VITPE_WAITVM(AI + WI);
        else VITPE_WAITVM(AI - 1 + WI);
      }
    } else if (AHEAD == 1 && kt == 0 && ep_pending != 0) {
      VITPE_WAITVM(EP_OPS);
    } else {
      VITPE_WAITVM(0);
    }
    asm volatile("s_barrier" ::: "memory");
  };
  auto prefetch = [&]() {                             // the first AHEAD stages of a tile -> buffers 0 .. AHEAD - 1
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
      if (i < nk) stage(i, i);
  };

  // ---- persistent over tiles: the next tile's first stages are issued BEFORE this tile's epilogue, so the epilogue's
  // arithmetic and stores run under the next tile's load latency.
  int m0, n0;
  tile_of(blockIdx.x, m0, n0);
  sources(m0, n0);
  prefetch();
  for (int id = blockIdx.x; id < total; id += gridDim.x) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    sync(0);
    if (AHEAD < nk) stage(AHEAD, AHEAD);
    // epilogue operands that do not depend on the product: issued now, used after the main loop
    // After the register exchange below a lane owns, for each of the two tile pairs p, 8 consecutive columns of row c:
    // columns 32 p + 16 (g & 1) + 8 (g >> 1) .. + 7 of the wave's 64.
    T* __restrict__ C = reinterpret_cast<T*>(a.C);
    const int gn = n0 + wn * 64 + 16 * (g & 1) + 8 * (g >> 1);
    const int row0 = m0 + wm * (16 * MT) + c;
    constexpr bool HAS_IN = EPI == EPI_BIAS_RESID || EPI == EPI_GELU_BWD;    // an [M,N] input of the epilogue (R or U)
    const T* __restrict__ IN = reinterpret_cast<const T*>(EPI == EPI_BIAS_RESID ? a.R : a.U);
    float bv[2][8];
    Chunk16 in_cur[2], in_nxt[2];
    auto load_in = [&](int mt, Chunk16* dst) {        // one pass AHEAD of its use: a wait for it never waits for a younger store
      int gm = row0 + 16 * mt;
      gm = gm < M ? gm : M - 1;
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[p] = *reinterpret_cast<const Chunk16*>(IN + (size_t)gm * N + gn + 32 * p);
    };
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i) bv[p][i] = (EPI != EPI_GELU_BWD && a.bias != nullptr) ? a.bias[gn + 32 * p + i] : 0.f;
    if (HAS_IN) load_in(0, in_cur);
    rd(0, 0, fwP, faP);
    rd(0, 1, fwQ, faQ);
    mm(fwP, faP);
    int buf = 1;                                      // NBUF >= 2
    for (int kt = 1; kt < nk; ++kt) {
      sync(kt);
      if (kt + AHEAD < nk) stage(kt + AHEAD, buf == 0 ? NBUF - 1 : buf - 1);   // (kt + AHEAD) % NBUF == (buf - 1) mod NBUF
      __builtin_amdgcn_sched_barrier(0);
      rd(buf, 0, fwP, faP);
      mm(fwQ, faQ);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      rd(buf, 1, fwQ, faQ);
      mm(fwP, faP);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    mm(fwQ, faQ);
    __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("s_barrier" ::: "memory");           // every fragment read of this tile is done: all stage buffers are free

    const int cm0 = m0;
    // ---- epilogue.
    // Two copies behind one wave-uniform branch: the common one has NO per-row conditions (a divergent `if` around a load
    // or store makes the compiler wait vmcnt(0) at the join -- that would drain the next tile's DMA and this tile's
    // stores at every pass); the ragged one (the wave's rows cross M) masks rows and ends with vmcnt(0).
    // the bias values and the first pass's input rows have been in flight since the tile's start; retire them BEFORE the
    // next tile's DMA is issued (behind it, the compiler's wait for them would be a wait for the DMA as well)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(bv[p][i]));
    if (HAS_IN) asm volatile("" : "+v"(in_cur[0]), "+v"(in_cur[1]));
    if (id + (int)gridDim.x < total) {
      tile_of(id + gridDim.x, m0, n0);
      sources(m0, n0);
      prefetch();
    }
    auto passes = [&](auto ragged_tag) {
      constexpr bool RAGGED = decltype(ragged_tag)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (HAS_IN && mt + 1 < MT) load_in(mt + 1, in_nxt);
        const int gm = row0 + 16 * mt;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          // lane (c, g) holds columns 16 nt + 4 g .. + 3 of row c for nt = 2p, 2p + 1; one v_permlane16_swap per dword
          // hands the odd lane group's tile-2p values to the even group and the even group's tile-(2p+1) values to the odd
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * p][mt][j]), __float_as_uint(acc[2 * p + 1][mt][j]),
                                                            false, false);
            v[j] = __uint_as_float(r[0]) + bv[p][j];
            v[4 + j] = __uint_as_float(r[1]) + bv[p][4 + j];
          }
          const size_t off = (size_t)gm * N + gn + 32 * p;
          if (EPI == EPI_BIAS_RESID) {
            float rv[8];
            chunk_to_f32<T>(in_cur[p], rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += rv[i];
          }
          if (EPI == EPI_BIAS_GELU) {
            const Chunk16 uo = f32_to_chunk<T>(v);
            if (!RAGGED || gm < M) *reinterpret_cast<Chunk16*>(reinterpret_cast<T*>(a.U) + off) = uo;
            gelu_erf_x8(v);
          }
          if (EPI == EPI_GELU_BWD) {
            float uv[8];
            chunk_to_f32<T>(in_cur[p], uv);
            gelu_erf_grad_mul_x8(v, uv);
          }
          const Chunk16 co = f32_to_chunk<T>(v);
          if (!RAGGED || gm < M) *reinterpret_cast<Chunk16*>(C + off) = co;
        }
        if (HAS_IN) { in_cur[0] = in_nxt[0]; in_cur[1] = in_nxt[1]; }
      }
    };
    if (cm0 + wm * (16 * MT) + 16 * MT <= M) {
      passes(std::false_type{});
      ep_pending = EP_OPS;
    } else {
      passes(std::true_type{});
      VITPE_WAITVM(0);
      ep_pending = 0;
    }
  }
#undef VITPE_WAITVM
}

static int gemm2d_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VITPE_GEMM2D");
    v = (e != nullptr && e[0] == '0') ? 0 : 1;
  }
  return v;
}

// bf16, no patch epilogue, K in whole 64-element steps, N in whole 256-column tiles, enough rows to fill the chip
bool gemm2d_takes(int dtype, int epi, int M, int N, int K) {
  return gemm2d_enabled() && dtype == 1 && (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID || epi == EPI_GELU_BWD) &&
         K % 64 == 0 && K >= 192 && N % 256 == 0 && M >= 2048 && (long long)M * N >= 4LL * 1024 * 1024;
}

static int g_forced_mt = 0;   // 0: heuristic; 4/5/6: forced (vitpe_debug_set_gemm2d_mt, tests and tools/kb_gemm2d.py)

static int gemm2d_pick_mt(int M, int N) {
  const int forced = g_forced_mt;
  if (forced == 4 || forced == 5 || forced == 6) return forced;
  const int cand[3] = {6, 5, 4};
  int best = 5;
  long long cost = -1;
  for (int i = 0; i < 3; ++i) {
    const int mt = cand[i];
    const long long tiles = (long long)((M + 32 * mt - 1) / (32 * mt)) * (N / 256);
    const long long rounds = (tiles + 255) / 256;
    // critical path ~ rounds x (MT k-loop work + a fixed per-tile part: prologue, epilogue)
    const long long cst = rounds * (mt * 8 + 6);
    if (cost < 0 || cst < cost) { cost = cst; best = mt; }
  }
  return best;
}

template <int MT>
static int gemm2d_launch_mt(int epi, const GemmNTArgs& a, hipStream_t s) {
  const int tiles = ((a.M + 32 * MT - 1) / (32 * MT)) * (a.N / 256);
  dim3 grid(tiles < 256 ? tiles : 256), block(512);   // persistent: one workgroup per CU
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm2d_kernel<MT, EPI_BIAS>), grid, block, 0, s, a); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm2d_kernel<MT, EPI_BIAS_GELU>), grid, block, 0, s, a); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm2d_kernel<MT, EPI_BIAS_RESID>), grid, block, 0, s, a); break;
    case EPI_GELU_BWD: hipLaunchKernelGGL((gemm2d_kernel<MT, EPI_GELU_BWD>), grid, block, 0, s, a); break;
    default: return (int)hipErrorInvalidValue;
  }
  VITPE_CHECK_LAUNCH();
}

int gemm2d_launch(int epi, const GemmNTArgs& a, hipStream_t s) {
  switch (gemm2d_pick_mt(a.M, a.N)) {
    case 4: return gemm2d_launch_mt<4>(epi, a, s);
    case 6: return gemm2d_launch_mt<6>(epi, a, s);
    default: return gemm2d_launch_mt<5>(epi, a, s);
  }
}

}  // namespace vitpe

// tests / A-B only (include/vitpe_debug.h): 0 = the host heuristic, 4 / 5 / 6 = that tile height for every launch
extern "C" int vitpe_debug_set_gemm2d_mt(int mt) {
  if (!(mt == 0 || mt == 4 || mt == 5 || mt == 6)) return (int)hipErrorInvalidValue;
  vitpe::g_forced_mt = mt;
  return 0;
}
