// Grouped weight-gradient GEMMs:  dW_p[n][k] += sum_m dY_p[m][n] X_p[m][k]   (+ dbias_p[n] += sum_m dY_p[m][n])
// for a whole list of problems p in ONE launch (the reference gets these from autograd's addmm
// backward, one ATen call per nn.Linear: vit.py:35,37, timm Mlp fc1/fc2).
//
// Why grouped: a ViT-tiny weight gradient has a tiny output (192x192 .. 768x192) and a huge
// contraction (M = batch x tokens = 33 280 rows), so every workgroup that shares an output block
// must add its partial result with fp32 atomics, and the atomics' traffic is (#workgroups per
// block) x (output bytes).  One launch per GEMM needs ~20-60 slices per block to fill 256 CUs;
// all 24 GEMMs of the model in one launch need ~3.5.  The activations / gradients stay resident
// (288 GB of HBM), so nothing forces the weight gradients to run inside the backward chain.
//
// Work decomposition: the unit of work is (output block of 192x192, stage of RPS token rows).
//  * stream-K: all units of all problems form one sequence, cut into equal contiguous runs, one per
//    workgroup (= one per CU): perfect balance, a workgroup flushes at most two partial blocks;
//  * table mode (big problem lists, e.g. the whole model): every block is cut into R row ranges
//    (R = 2 x CUs / blocks, two rounds of workgroups) and the blocks of one problem over the same
//    range -- which read the same X (or dY) rows -- are placed on the same XCD at the same time, so
//    those rows come out of that XCD's L2 instead of HBM (measured 293 -> 285 us for 24 GEMMs).
//
// Block 192 (dY columns) x 192 (X columns), 12 waves as 4 x 3, wave tile 48 x 64 (12 MFMAs per
// 7 transposed operand fragments and K32 chunk).  A stage is staged row-major as it lies in HBM
// (16-B coalesced loads, register prefetch one stage ahead, double-buffered LDS, one barrier per
// stage) and consumed through transposed LDS reads (ds_read_b64_tr_b16).  bf16 rows are 384 B
// (no padding); the 32-B pieces of a row are XOR-swizzled so that the 16 row-pieces one transposed
// read touches per 32 lanes cover all 64 banks exactly once.
#include "common.h"
#include <stdlib.h>

namespace vitpe {

constexpr int WG_MAXPROB = 28;   // the argument block (descriptors + placement table) must stay below the 4 KB kernarg limit
constexpr int WG_BLK = 192;

struct WgProb {
  const void* dY;  // [M,N] T
  const void* X;   // [M,K] T
  float* dW;       // [N,K] fp32, accumulated into
  float* dbias;    // [N] fp32 or null, accumulated into
  const float* xmean;   // x_op == LayerNorm: row statistics [M] and affine parameters [K] of the X operand
  const float* xrstd;
  const float* xgamma;
  const float* xbeta;
  int M, N, K;
  int xop;
  int unit0;       // index of this problem's first work unit
  int nbn;         // 192-wide blocks along N
  int stages;      // ceil(M / RPS)
};
constexpr int WG_TABLE = 512;
struct WgArgs {
  int nprob, total_units, units_per_wg;
  int use_table;   // 1: workgroup i runs table[i] = (problem << 11 | block << 5 | row range), 0xFFFF = idle; 2: window mode
  int nranges;     // row ranges per block in table mode (window mode: of the blocks of the last, partial window)
  int full_rounds, total_blocks;   // window mode (use_table == 2): whole windows of gridDim.x blocks; blocks in the list
  int tail_rounds;                 // ... and the windows its last, partial window's (row range, block) pairs are dealt over
  int bk;          // block width along K: 192 (wgrad_group_kernel) or 384 (wgrad_wide_kernel)
  int range_major; // window mode, lists below two windows: EVERY block is cut into nranges row ranges, the (range, block) pairs
                   // are dealt range-major in full_rounds windows (73 blocks x 7 ranges = 511 of 512 slots: two balanced rounds)
  unsigned long long* census;   // CENSUS build only (vitpe_debug_wgrad_census): s_memtime stamps
  WgProb p[WG_MAXPROB];
  unsigned short table[WG_TABLE];
};
static_assert(sizeof(WgArgs) <= 4096, "kernel argument block");
constexpr int WG_CENSUS_STAGES = 24, WG_CENSUS_SLOTS = 4;

template <typename T> struct WgLayout;
template <> struct WgLayout<bf16> {
  static constexpr int LD = WG_BLK;
  // element offset of 16-B chunk cc (8 elements) of row r: 32-B pieces swizzled within groups of four
  static __device__ __forceinline__ int chunk(int r, int cc) {
    const int s = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return r * LD + ((((cc >> 1) ^ s)) << 4) + ((cc & 1) << 3);
  }
  // transposed fragment: rows rb0..rb0+3 / rb1..rb1+3 (rb multiple of 4), 16 columns at c0 (multiple of 16)
  static __device__ __forceinline__ Frag<bf16> tr(const bf16* tile, int rb0, int rb1, int c0) { return tr_ld<LD>(tile, rb0, rb1, c0); }
  template <int LD>
  static __device__ __forceinline__ Frag<bf16> tr_ld(const bf16* tile, int rb0, int rb1, int c0) {
    const int i = threadIdx.x & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int r0 = rb0 + q, r1 = rb1 + q;
    // rows of 384 B alternate between the two halves of the banks by themselves; rows that are a multiple of 256 B (LD = 384
    // elements) all start on bank 0: the swizzle then also flips the 128-B half with the row's parity (PMC on the first
    // version of the wide kernel: 33 % of the LDS cycles were bank conflicts)
    constexpr int PAR = (LD % 128 == 0) ? 4 : 0;
    const int s0 = (((r0 >> 1) & 1) | (((r0 >> 3) & 1) << 1)) ^ ((r0 & 1) * PAR), s1 = (((r1 >> 1) & 1) | (((r1 >> 3) & 1) << 1)) ^ ((r1 & 1) * PAR);
    const bf16* a0 = tile + r0 * LD + (((c0 >> 4) ^ s0) << 4) + 4 * p;
    const bf16* a1 = tile + r1 * LD + (((c0 >> 4) ^ s1) << 4) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    Frag<bf16> f;
    f.v = __builtin_bit_cast(bf16x8, both);
    return f;
  }
};
template <> struct WgLayout<float> {  // exact-fp32 parity mode: padded rows, scalar column gathers
  static constexpr int LD = WG_BLK + 4;
  static __device__ __forceinline__ int chunk(int r, int cc) { return r * LD + cc * 4; }
  static __device__ __forceinline__ Frag<float> tr(const float* tile, int rb0, int rb1, int c0) {
    return ld_frag_tr(tile, LD, rb0, rb1, c0);
  }
};

// Work assignment shared by the two kernels: calls run(u0, uend) for every contiguous run of work units (one unit = one
// (output block, 64-row stage)) this workgroup owns.
template <class RunFn>
VITPE_DEV void wg_dispatch(const WgArgs& a, RunFn run) {
  // Window mode (big lists: at least two windows of gridDim.x blocks): every workgroup takes WHOLE blocks, one per round;
  // round r runs the gridDim.x consecutive blocks of window r at the same time and gives each XCD (workgroup id % 8) a
  // contiguous eighth of them -- with the n-blocks of a problem adjacent, the workgroups of an XCD then walk the same token
  // rows of the same dY / X column blocks in step and the second reader is served by that XCD's L2.  (Stream-K hands
  // every CU a contiguous run of units instead: the blocks alive at one time are far apart in the list, nothing is shared
  // and every stage of every block comes through the fabric -- 11.4 GB per launch on the ViT-B/16 list for 1.9 GB of
  // operands.)  One flush per block; the last, partial window is cut into row ranges so that it still fills the chip.
  if (a.use_table == 2) {
    const int G = gridDim.x, idx = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int nrounds = a.range_major ? a.full_rounds : a.full_rounds + a.tail_rounds;
    for (int r = 0; r < nrounds; ++r) {
      int gb, rng = 0, R = 1;
      if (a.range_major) {
        const int vb = r * G + idx;
        R = a.nranges;
        rng = vb / a.total_blocks;
        gb = rng < R ? vb - rng * a.total_blocks : a.total_blocks;
      } else if (r < a.full_rounds) {
        gb = r * G + idx;
      } else {
        // the blocks left over by the whole windows, each cut into R row ranges; the (range, block) pairs range-major over
        // tail_rounds windows (adjacent workgroups = adjacent blocks over the same rows, as in a whole window)
        R = a.nranges;
        const int rem = a.total_blocks - a.full_rounds * G, q = (r - a.full_rounds) * G + idx;
        if (q >= rem * R) continue;
        rng = q / rem;
        gb = a.full_rounds * G + q % rem;
      }
      if (gb >= a.total_blocks) continue;
      int pi = 0, b0 = 0;
      for (int i = 0; i < a.nprob; ++i) {
        const int nb = a.p[i].nbn * ((a.p[i].K + a.bk - 1) / a.bk);
        if (gb >= b0 + nb) { b0 += nb; pi = i + 1; } else break;
      }
      const WgProb& P = a.p[pi];
      const int spr = (P.stages + R - 1) / R;
      const int ub = P.unit0 + (gb - b0) * P.stages;
      const int u0 = ub + min(P.stages, rng * spr), uend = ub + min(P.stages, (rng + 1) * spr);
      if (u0 < uend) run(u0, uend);
    }
    return;
  }
  int u0, uend;
  if (a.use_table == 1) {
    const unsigned ent = a.table[blockIdx.x];
    if (ent == 0xFFFFu) return;
    const WgProb& P = a.p[ent >> 11];
    const int blk = (int)((ent >> 5) & 0x3Fu), rng = (int)(ent & 0x1Fu);
    const int spr = (P.stages + a.nranges - 1) / a.nranges;
    u0 = P.unit0 + blk * P.stages + min(P.stages, rng * spr);
    uend = P.unit0 + blk * P.stages + min(P.stages, (rng + 1) * spr);
  } else {
    u0 = blockIdx.x * a.units_per_wg;
    uend = min(a.total_units, u0 + a.units_per_wg);
  }
  if (u0 >= uend) return;
  run(u0, uend);
}

// LNX: compiled with the LayerNorm-operand path (x_op).  A separate instantiation: its two statistics loads per staged
// X chunk double the VMEM instruction count of a stage, and this kernel lives on the vector-memory path (measured:
// 288 -> 342 us for a list WITHOUT any LayerNorm problem when the loads were unconditional).
template <typename T, bool CENSUS, bool LNX>
__global__ __launch_bounds__(768) void wgrad_group_kernel(WgArgs a) {
  using LY = WgLayout<T>;
  constexpr int RPS = 128 / (int)sizeof(T);  // token rows per stage: 64 bf16 / 32 fp32
  constexpr int CPS = RPS / 32;              // K32 chunks per stage
  constexpr int CHN = CH<T>::n;
  constexpr int CPR = WG_BLK / CHN;          // 16-B chunks per slab row
  static_assert(RPS * CPR == 2 * 768, "a slab is two chunks per thread");
  constexpr int SLAB = RPS * LY::LD;
  __shared__ __attribute__((aligned(16))) T sm[2 * 2 * SLAB];  // [buffer][Y | X][RPS][LD]

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / 3, wk = wave % 3;

  struct Cur {
    const T* dY; const T* X; float* dW; float* dbias;
    const float* xmean; const float* xrstd; const float* xgamma; const float* xbeta;
    int M, N, K, n0, k0, stage, stages, xop;
  };
  auto decode = [&](int u, Cur& s) {
    int pi = 0;
    for (int i = 1; i < a.nprob; ++i) pi = (u >= a.p[i].unit0) ? i : pi;
    const WgProb& P = a.p[pi];
    const int ub = u - P.unit0, blk = ub / P.stages;
    s.dY = reinterpret_cast<const T*>(P.dY); s.X = reinterpret_cast<const T*>(P.X); s.dW = P.dW; s.dbias = P.dbias;
    s.M = P.M; s.N = P.N; s.K = P.K; s.stages = P.stages;
    // (plain problems: a dummy word of their own dY -- read-only and cache-resident; NOT dW, whose lines are being
    //  updated by other workgroups' flush atomics at the memory side and would miss every time)
    s.xmean = P.xop ? P.xmean : reinterpret_cast<const float*>(P.dY);
    s.xrstd = P.xop ? P.xrstd : reinterpret_cast<const float*>(P.dY);
    s.xgamma = P.xgamma; s.xbeta = P.xbeta; s.xop = P.xop;
    s.stage = ub - blk * P.stages;
    s.n0 = (blk % P.nbn) * WG_BLK;
    s.k0 = (blk / P.nbn) * a.bk;
  };

  const Chunk16 zero = {0u, 0u, 0u, 0u};
  Chunk16 rg[4];
  // X operand = LayerNorm(X) (x_op): the staged rows become xhat = (x - mean) * rstd here (two scalars per row);
  // gamma and beta are applied to the finished block, dW[n][k] += gamma[k] * (dY^T xhat)[n][k] + beta[k] * colsum(dY)[n]
  // (the column sums come off the matrix core like the bias gradient) -- one fma per staged element, fp32 affine part.
  float lnA[2], lnB[2];   // xhat = x * lnA + lnB per staged X chunk
  bool ln_staged = false;
  auto gload = [&](const Cur& s) {
    const int mb = s.stage * RPS;
    ln_staged = LNX && s.xop == 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 768 * (i & 1), row = q / CPR, cc = q % CPR;
      const int gm = mb + row;
      if (i < 2) {
        const int gn = s.n0 + cc * CHN;
        rg[i] = (gm < s.M && gn < s.N) ? *reinterpret_cast<const Chunk16*>(s.dY + (size_t)gm * s.N + gn) : zero;
      } else {
        const int gk = s.k0 + cc * CHN;
        const bool ok = gm < s.M && gk < s.K;
        rg[i] = ok ? *reinterpret_cast<const Chunk16*>(s.X + (size_t)gm * s.K + gk) : zero;
        // row statistics: UNCONDITIONAL loads (plain problems read a dummy word of their own dW): a branch around them
        // made the compiler drain the whole prefetch (vmcnt(0)) before it, i.e. every stage waited out its HBM latency
        if (LNX) {
          // (only LOADED here, like the chunks: consuming them now would wait for every load of the burst; unconditional --
          //  plain problems of a mixed list read a dummy word -- because a branch made the compiler drain the prefetch)
          const int gs = (s.xop == 1) ? min(gm, s.M - 1) : 0;
          lnA[i - 2] = s.xrstd[gs];
          lnB[i - 2] = s.xmean[gs];
        }
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 768 * (i & 1), row = q / CPR, cc = q % CPR;
      Chunk16 v = rg[i];
      if (LNX && i >= 2 && ln_staged) {
        float f[CHN];
        chunk_to_f32<T>(v, f);
        // (rows past M become -mean * rstd instead of 0: harmless, their dY rows are zero)
        const float rs = lnA[i - 2], nb = -lnB[i - 2] * rs;
#pragma unroll
        for (int t = 0; t < CHN; ++t) f[t] = fmaf(f[t], rs, nb);
        v = f32_to_chunk<T>(f);
      }
      *reinterpret_cast<Chunk16*>(sm + (buf * 2 + (i >> 1)) * SLAB + LY::chunk(row, cc)) = v;
    }
  };

  f32x4 acc[3][4], accb[3];
  auto clear = [&]() {
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      accb[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  Frag<T> ones;
#pragma unroll
  for (int t = 0; t < 8; ++t) ones.v[t] = from_f32<T>(1.0f);

  auto compute = [&](int buf, bool bias) {
    const T* sY = sm + (buf * 2 + 0) * SLAB;
    const T* sX = sm + (buf * 2 + 1) * SLAB;
#pragma unroll
    for (int cs = 0; cs < CPS; ++cs) {
      const int rb0 = cs * 32 + 8 * g, rb1 = rb0 + 4;
      Frag<T> fy[3], fx[4];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) fy[nt] = LY::tr(sY, rb0, rb1, wn * 48 + 16 * nt);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) fx[kt] = LY::tr(sX, rb0, rb1, wk * 64 + 16 * kt);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) mma(fy[nt], fx[kt], acc[nt][kt]);
      if (bias) {  // column sums of dY on the matrix core: dY^T . 1
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) mma(fy[nt], ones, accb[nt]);
      }
    }
  };
  auto flush = [&](const Cur& s, bool bias) {
    float gam[4], bet[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int gk = s.k0 + wk * 64 + 16 * kt + c;
      gam[kt] = (LNX && s.xop == 1 && gk < s.K) ? s.xgamma[gk] : 1.f;
      bet[kt] = (LNX && s.xop == 1 && gk < s.K) ? s.xbeta[gk] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gn = s.n0 + wn * 48 + 16 * nt + 4 * g + r;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int gk = s.k0 + wk * 64 + 16 * kt + c;
          float v = acc[nt][kt][r];
          if (LNX && s.xop == 1) v = fmaf(v, gam[kt], bet[kt] * accb[nt][r]);
          if (gn < s.N && gk < s.K) atomicAdd(s.dW + (size_t)gn * s.K + gk, v);
        }
        if (bias && c == 0 && gn < s.N) atomicAdd(s.dbias + gn, accb[nt][r]);
      }
    clear();
  };

  // Work assignment.  Stream-K (default fallback): an equal contiguous run of the flattened unit sequence.
  // Table mode: one (block, row range) per workgroup, placed by the host so that the blocks of one problem that
  // share an operand over the same rows (e.g. the four n-blocks of fc1 all read xn2) run at the same time on
  // the SAME XCD (workgroup id % 8) -- the shared rows are fetched from HBM once and hit in that XCD's L2.
  bool first_run = true;
  auto run = [&](int u0, int uend) {
    if (!first_run) __syncthreads();   // the previous run's last stage may still be read by a slower wave
    first_run = false;
  Cur cur, nxt;
    decode(u0, cur);
    gload(cur);
    clear();
    // CENSUS: lane 0 of every wave stamps the shader clock of the first 24 stages: 0 stage stored, 2 next loads
    // issued, 1 barrier passed, 3 MFMAs done
    auto stamp = [&](int u, int slot) {
      if (CENSUS && lane == 0 && u - u0 < WG_CENSUS_STAGES)
        a.census[(((size_t)blockIdx.x * 12 + wave) * WG_CENSUS_STAGES + (u - u0)) * WG_CENSUS_SLOTS + slot] = __builtin_amdgcn_s_memtime();
    };
    for (int u = u0; u < uend; ++u) {
      const int buf = (u - u0) & 1;
      sstore(buf);
      stamp(u, 0);
      nxt = cur;
      bool flush_now = (u + 1 == uend);
      if (u + 1 < uend) {
        if (cur.stage + 1 < cur.stages) nxt.stage = cur.stage + 1;
        else { decode(u + 1, nxt); flush_now = true; }
        gload(nxt);  // in flight under this stage's MFMAs
      }
      stamp(u, 2);
      __syncthreads();
      stamp(u, 1);
      const bool bias = (cur.dbias != nullptr) && cur.k0 == 0 && wk == 0;
      compute(buf, bias || (LNX && cur.xop == 1));   // LayerNorm operand: every wave needs the column sums of dY (beta term)
      stamp(u, 3);
      if (flush_now) flush(cur, bias);
      cur = nxt;
    }
  };

  wg_dispatch(a, run);
}

// =========================================================================================
// Wide blocks: 192 (dY columns) x 384 (X columns) per workgroup, 8 waves as 2 x 4 with 96 x 96 wave tiles, for lists whose
// every problem has N % 192 == 0 and K % 384 == 0 (ViT-B/16: K in {768, 3072}).  Why: at d = 768 the operands are served
// by the L2 (every X column block is re-read by N / 192 output blocks) and the 192 x 192 kernel is bound by its own LDS
// traffic -- a 48 x 64 wave tile reads (48 + 64) / (48 x 64) fragment bytes per MFMA flop, 1.7 K cycles of LDS
// bandwidth per stage against 1.15 K cycles of MFMAs; a 96 x 96 wave tile reads 43 % less per flop and a block re-reads
// dY half as often.  No staging registers (the accumulators take 144 + 24 of the 256): a stage (64 rows of both operands,
// 72 KB) comes in by LDS-DMA, 9 instructions per wave, with the bank swizzle of the 192-wide kernel applied on the SOURCE
// side (LDS position p of a slab holds source chunk swz(p): an LDS-DMA instruction writes 64 consecutive 16-B slots).
// bf16 only, no LayerNorm operand (lists that need either stay on wgrad_group_kernel).
// =========================================================================================
constexpr int WW_BK = 384, WW_TH = 512;

__global__ __launch_bounds__(WW_TH) void wgrad_wide_kernel(WgArgs a) {
  using T = bf16;
  using LY = WgLayout<bf16>;
  constexpr int RPS = 64, YLD = WG_BLK, XLD = WW_BK;
  constexpr int YSLAB = RPS * YLD, XSLAB = RPS * XLD, STG = YSLAB + XSLAB;   // elements per stage: 72 KB
  constexpr int YI = YSLAB / 512, XI = XSLAB / 512, NI = (YI + XI) / 8;      // LDS-DMA instructions: 24 + 48, 9 per wave
  static_assert((YI + XI) % 8 == 0, "instructions per wave");
  __shared__ __attribute__((aligned(1024))) T sm[2 * STG];

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;

  struct Cur { const T* dY; const T* X; float* dW; float* dbias; int M, N, K, n0, k0, stage, stages; };
  auto decode = [&](int u, Cur& s) {
    int pi = 0;
    for (int i = 1; i < a.nprob; ++i) pi = (u >= a.p[i].unit0) ? i : pi;
    const WgProb& P = a.p[pi];
    const int ub = u - P.unit0, blk = ub / P.stages;
    s.dY = reinterpret_cast<const T*>(P.dY); s.X = reinterpret_cast<const T*>(P.X); s.dW = P.dW; s.dbias = P.dbias;
    s.M = P.M; s.N = P.N; s.K = P.K; s.stages = P.stages;
    s.stage = ub - blk * P.stages;
    s.n0 = (blk % P.nbn) * WG_BLK;
    s.k0 = (blk / P.nbn) * WW_BK;
  };
  // source chunk of LDS slot (row r, 16-B slot cl) of a slab: the swizzle of WgLayout<bf16>::chunk is an involution
  auto swz = [](int r, int cl) { return ((((cl >> 1) ^ (((r >> 1) & 1) | (((r >> 3) & 1) << 1))) << 1) | (cl & 1)); };
  auto swzx = [](int r, int cl) {   // X slab (768-B rows): + the row-parity flip of the 128-B half (WgLayout<bf16>::tr_ld)
    return ((((cl >> 1) ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) ^ ((r & 1) << 2))) << 1) | (cl & 1));
  };
  static_assert(YI == 24 && XI == 48, "instruction j of a wave: j < 3 -> dY slab, else X slab");
  const unsigned sm_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)sm;
  auto dma = [&](const Cur& s, int buf) {
    const int mb = s.stage * RPS;
    // (the slot -> (row, chunk) divisions are loop-invariant: hoisted, their 18 results lived through the stage loop and
    //  were spilled -- and a scratch reload between two LDS-DMA instructions is a vmcnt(0) wait.  Recomputed per stage.)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int q = wave + 8 * j;              // wave-uniform
      const T* src;
      if (j < 3) {
        const int p = q * 64 + ln, r = p / (YLD / 8), cl = p % (YLD / 8);
        src = s.dY + (size_t)min(mb + r, s.M - 1) * s.N + s.n0 + swz(r, cl) * 8;
      } else {
        const int p = (q - YI) * 64 + ln, r = p / (XLD / 8), cl = p % (XLD / 8);
        src = s.X + (size_t)min(mb + r, s.M - 1) * s.K + s.k0 + swzx(r, cl) * 8;
      }
      // (inline asm: behind the builtin hipcc puts an s_waitcnt vmcnt(0) before the next read of `sm` -- the prefetch of
      //  stage u + 1 would be waited out before stage u's first MFMA.  The waits are placed by hand in run().)
      const unsigned dst = sm_lds + (unsigned)(buf * STG + q * 512) * 2;      // LDS byte address, wave-uniform
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  };

  // bias gradient (column sums of dY, off the matrix core like the 192-wide kernel): the six n tiles of a wave row are
  // shared out over its waves wk = 0, 1, 2 (two tiles each) -- six tiles on wave 0 alone were 24 registers nobody has
  f32x4 acc[6][6], accb[2];
  auto clear = [&]() {
    accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
      for (int kt = 0; kt < 6; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  Frag<T> ones;
#pragma unroll
  for (int t = 0; t < 8; ++t) ones.v[t] = from_f32<T>(1.0f);

  auto compute = [&](int buf, bool bias) {
    const T* sY = sm + buf * STG;
    const T* sX = sY + YSLAB;
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
      const int rb0 = cs * 32 + 8 * g, rb1 = rb0 + 4;
      __builtin_amdgcn_sched_barrier(0);       // (fragments of one K32 chunk at a time: 9 live, not 18)
      Frag<T> fy[6];
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) fy[nt] = LY::tr_ld<YLD>(sY, rb0, rb1, wn * 96 + 16 * nt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        Frag<T> fx[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) fx[t] = LY::tr_ld<XLD>(sX, rb0, rb1, wk * 96 + 16 * (3 * h + t));
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
          for (int t = 0; t < 3; ++t) mma(fy[nt], fx[t], acc[nt][3 * h + t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (bias) {  // column sums of dY on the matrix core: dY^T . 1 (this wave's two tiles: wave-uniform choice)
        if (wk == 0) { mma(fy[0], ones, accb[0]); mma(fy[1], ones, accb[1]); }
        else if (wk == 1) { mma(fy[2], ones, accb[0]); mma(fy[3], ones, accb[1]); }
        else { mma(fy[4], ones, accb[0]); mma(fy[5], ones, accb[1]); }
      }
    }
  };
  auto flush = [&](const Cur& s, bool bias) {
#pragma unroll
    for (int nt = 0; nt < 6; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gn = s.n0 + wn * 96 + 16 * nt + 4 * g + r;
        float* row = s.dW + (size_t)gn * s.K + s.k0 + wk * 96 + c;
#pragma unroll
        for (int kt = 0; kt < 6; ++kt) atomicAdd(row + 16 * kt, acc[nt][kt][r]);
      }
    if (bias && c == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(s.dbias + s.n0 + wn * 96 + 16 * (2 * wk + t) + 4 * g + r, accb[t][r]);
    }
    clear();
  };

  bool first_run = true;
  auto run = [&](int u0, int uend) {
    if (!first_run) __syncthreads();   // the previous run's last stage may still be read by a slower wave
    first_run = false;
    Cur cur, nxt;
    decode(u0, cur);
    dma(cur, 0);
    clear();
    for (int u = u0; u < uend; ++u) {
      const int buf = (u - u0) & 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces of stage u (and any flush before them)
      __syncthreads();                         // everybody's pieces; everybody has left the other buffer
      if (cur.stage * RPS + RPS > cur.M) {     // the block's last, partial stage: dY rows past M contribute nothing
        const int first = cur.M - cur.stage * RPS;   // rows [first, 64) of the dY slab
        T* sY = sm + buf * STG;
        for (int q = tid; q < (RPS - first) * (YLD / 8); q += WW_TH)
          *reinterpret_cast<Chunk16*>(sY + first * YLD + q * 8) = (Chunk16){0u, 0u, 0u, 0u};
        __syncthreads();
      }
      nxt = cur;
      bool flush_now = (u + 1 == uend);
      if (u + 1 < uend) {
        if (cur.stage + 1 < cur.stages) nxt.stage = cur.stage + 1;
        else { decode(u + 1, nxt); flush_now = true; }
        dma(nxt, buf ^ 1);                     // in flight under this stage's MFMAs
      }
      const bool bias = (cur.dbias != nullptr) && cur.k0 == 0 && wk < 3;
      compute(buf, bias);
      if (flush_now) flush(cur, bias);
      cur = nxt;
    }
  };
  wg_dispatch(a, run);
}

}  // namespace vitpe

using namespace vitpe;

struct vitpe_wgrad_problem_abi {  // mirrors include/vitpe.h: vitpe_wgrad_problem
  const void* dY;
  const void* X;
  float* dW;
  float* dbias;
  int M, N, K, x_op;
  const float* x_mean;
  const float* x_rstd;
  const float* x_gamma;
  const float* x_beta;
};

static int wgrad_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

static int wgrad_group_launch(int dtype, const void* problems, int nprob, unsigned long long* census, hipStream_t stream);
// On by default.  (It first ran SLOWER inside the ViT-B/16 step than stand-alone: the single-GPU step hands over all 49
// problems as launches of 28 + 21, and the placement below -- tuned for 192 x 192 blocks -- sent the 488 wide blocks of the
// second launch to stream-K and left the first one a 62 %-full last window.  With equal launches (25 + 24) and the tail of
// a window list dealt as (row range, block) pairs over several windows: step 11.30 -> 11.11 ms.)
// vitpe_debug_set_wgrad_wide(0): eligible lists stay on the 192 x 192 kernel (A/B, tests).
static bool g_wide_ok = true;
extern "C" int vitpe_debug_set_wgrad_wide(int on) { g_wide_ok = on != 0; return 0; }

extern "C" int vitpe_wgrad_group(int dtype, const void* problems, int nprob, hipStream_t stream) {
  return wgrad_group_launch(dtype, problems, nprob, nullptr, stream);
}

// debug: the same launch (bf16) with per-wave s_memtime stamps of the first 24 stages of every workgroup:
// census[((wg * 12 + wave) * 24 + stage) * 4 + slot], >= grid * 12 * 24 * 4 entries (tools/census_wgrad.py)
extern "C" int vitpe_debug_wgrad_census(int dtype, const void* problems, int nprob, unsigned long long* census,
                                        hipStream_t stream) {
  VITPE_REQUIRE(census != nullptr && dtype == 1);
  return wgrad_group_launch(dtype, problems, nprob, census, stream);
}

static int wgrad_group_launch(int dtype, const void* problems, int nprob, unsigned long long* census, hipStream_t stream) {
  VITPE_REQUIRE(problems && nprob >= 0 && nprob <= WG_MAXPROB && (dtype == 0 || dtype == 1));
  const vitpe_wgrad_problem_abi* pr = reinterpret_cast<const vitpe_wgrad_problem_abi*>(problems);
  const int RPS = dtype == 1 ? 64 : 32, CHN = dtype == 1 ? 8 : 4;
  WgArgs a{};
  a.census = census;
  // 192 x 384 blocks (wgrad_wide_kernel) when every problem allows them: bf16, plain X operand, whole blocks
  bool wide = g_wide_ok && dtype == 1 && census == nullptr && nprob > 0;
  for (int i = 0; i < nprob; ++i)
    wide = wide && (pr[i].M == 0 || (pr[i].x_op == 0 && pr[i].N % WG_BLK == 0 && pr[i].K % WW_BK == 0));
  a.bk = wide ? WW_BK : WG_BLK;
  int units = 0, np = 0;
  for (int i = 0; i < nprob; ++i) {
    const vitpe_wgrad_problem_abi& p = pr[i];
    VITPE_REQUIRE(p.dY && p.X && p.dW && p.M >= 0 && p.N > 0 && p.K > 0 && p.N % CHN == 0 && p.K % CHN == 0);
    if (p.M == 0) continue;
    WgProb& q = a.p[np++];
    VITPE_REQUIRE(p.x_op == 0 || (p.x_op == 1 && p.x_mean && p.x_rstd && p.x_gamma && p.x_beta));
    q.dY = p.dY; q.X = p.X; q.dW = p.dW; q.dbias = p.dbias; q.M = p.M; q.N = p.N; q.K = p.K;
    q.xop = p.x_op; q.xmean = p.x_mean; q.xrstd = p.x_rstd; q.xgamma = p.x_gamma; q.xbeta = p.x_beta;
    q.unit0 = units;
    q.nbn = (p.N + WG_BLK - 1) / WG_BLK;
    q.stages = (p.M + RPS - 1) / RPS;
    const long long nu = (long long)q.nbn * ((p.K + a.bk - 1) / a.bk) * q.stages;
    VITPE_REQUIRE(units + nu < (1LL << 30));
    units += (int)nu;
  }
  if (units == 0) return 0;
  a.nprob = np;
  a.total_units = units;
  const int ncu = wgrad_cu_count();
  int grid = 0;
  // ---- table mode: (block, row range) per workgroup, operand-sharing blocks co-located on one XCD -----------
  int total_blocks = 0, max_blocks = 0, min_stages = 1 << 30;
  for (int i = 0; i < np; ++i) {
    const int nb = a.p[i].nbn * ((a.p[i].K + a.bk - 1) / a.bk);
    total_blocks += nb;
    max_blocks = nb > max_blocks ? nb : max_blocks;
    min_stages = a.p[i].stages < min_stages ? a.p[i].stages : min_stages;
  }
  int R = (2 * ncu) / (total_blocks > 0 ? total_blocks : 1);   // two rounds of workgroups over the CUs
  if (R > 255) R = 255;
  // (many row ranges = many partial-block flushes: with R = 42 for a single layer the atomics cost more than
  // the shared reads save -- measured 91 vs 68 us -- so small problem lists stay on the stream-K path)
  // (the whole-model list -- 73 blocks -- does not fit the table with 7 row ranges and falls through to stream-K; forcing
  //  6 ranges so that it fits: 298 us against stream-K's 293.5, 5 ranges 319, 4 ranges 374 -- co-location cuts the HBM reads
  //  by a third but the kernel is bound by CU-side delivery, not by HBM (round-2 A/B).  A 576-slot table that holds 7
  //  ranges: 352 us -- some XCD then gets more than the 64 workgroups its 32 CUs take in two rounds)
  if (R >= 2 && R <= 8 && R <= min_stages / 4 && max_blocks <= 63 && np <= 31) {
    // groups = (problem, row range); greedy: next group to the XCD with the fewest workgroups so far
    int len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool fits = true;
    for (int i = 0; i < WG_TABLE; ++i) a.table[i] = 0xFFFFu;
    for (int r = 0; r < R && fits; ++r)
      for (int i = 0; i < np && fits; ++i) {
        const int nb = a.p[i].nbn * ((a.p[i].K + a.bk - 1) / a.bk);
        int x = 0;
        for (int k = 1; k < 8; ++k) x = len[k] < len[x] ? k : x;
        for (int b = 0; b < nb; ++b) {
          const int id = (len[x] + b) * 8 + x;
          if (id >= WG_TABLE) { fits = false; break; }
          a.table[id] = (unsigned short)(((unsigned)i << 11) | ((unsigned)b << 5) | (unsigned)r);
        }
        len[x] += nb;
      }
    if (fits) {
      int mx = 0;
      for (int k = 0; k < 8; ++k) mx = len[k] > mx ? len[k] : mx;
      a.use_table = 1;
      a.nranges = R;
      grid = mx * 8;
    }
  }
  const int G = ncu & ~7;
  if (!a.use_table && G >= 8 && total_blocks >= G) {
    a.use_table = 2;
    a.total_blocks = total_blocks;
    a.full_rounds = total_blocks / G;
    const int rem = total_blocks - a.full_rounds * G;
    // the leftover blocks in R row ranges over T = ceil(rem R / G) windows: the tail then lasts T / R of a block's time.
    // Smallest T / R wins, ties go to fewer ranges (fewer partial flushes).  (Round 2 allowed one tail window only, R =
    // G / rem: 160 leftover blocks of a 672-block list ran as ONE 62 %-full window a whole block long.)
    int bestR = 1, bestT = rem > 0 ? 1 : 0;
    int Rmax = min_stages / 4;
    if (Rmax > 8) Rmax = 8;         // (every range is one more partial flush of the block: fp32 atomics)
    for (int Rc = 2; rem > 0 && Rc <= Rmax; ++Rc) {
      const int Tc = (rem * Rc + G - 1) / G;
      if ((long long)Tc * bestR < (long long)bestT * Rc) { bestR = Rc; bestT = Tc; }
    }
    a.nranges = bestR;
    a.tail_rounds = bestT;
    grid = G;
  }
  // lists below two windows (the CIFAR model: 73 blocks): the same placement on (row range, block) pairs when some R fills
  // k rounds of the chip almost exactly -- co-location cuts the fetched bytes by a third (PMC: 1.77 -> 1.20 GB) as the
  // table mode does, without its second, 71 %-full round
  if (!a.use_table && G >= 8 && total_blocks > 0 && total_blocks < G) {
    int bestR = 0, bestk = 0;
    double besteff = 0.0;
    for (int k = 1; k <= 3; ++k) {
      int Rk = (k * G) / total_blocks;
      if (Rk > min_stages / 4) Rk = min_stages / 4;
      if (Rk > 31) Rk = 31;
      if (Rk < 2) continue;
      const double eff = (double)total_blocks * Rk / (double)(((total_blocks * Rk + G - 1) / G) * G);
      if (eff > besteff + 1e-9) { besteff = eff; bestR = Rk; bestk = k; }
    }
    if (bestR >= 2 && besteff >= 0.97) {
      a.use_table = 2;
      a.range_major = 1;
      a.total_blocks = total_blocks;
      a.nranges = bestR;
      a.full_rounds = (total_blocks * bestR + G - 1) / G;
      grid = G;
      (void)bestk;
    }
  }
  if (!a.use_table) {
    const int wgs = units < ncu ? units : ncu;
    a.units_per_wg = (units + wgs - 1) / wgs;
    grid = (units + a.units_per_wg - 1) / a.units_per_wg;
  }
  bool lnx = false;
  for (int i = 0; i < np; ++i) lnx = lnx || a.p[i].xop != 0;
  // (the census instantiation exists for bf16 operands without the LayerNorm X operand only: anything else would silently
  //  accumulate gradients of the wrong operands)
  if (census != nullptr) VITPE_REQUIRE(!lnx && dtype == 1);
  if (wide) hipLaunchKernelGGL(wgrad_wide_kernel, dim3(grid), dim3(WW_TH), 0, stream, a);
  else if (census != nullptr) hipLaunchKernelGGL((wgrad_group_kernel<bf16, true, false>), dim3(grid), dim3(768), 0, stream, a);
  else if (dtype == 1 && lnx) hipLaunchKernelGGL((wgrad_group_kernel<bf16, false, true>), dim3(grid), dim3(768), 0, stream, a);
  else if (dtype == 1) hipLaunchKernelGGL((wgrad_group_kernel<bf16, false, false>), dim3(grid), dim3(768), 0, stream, a);
  else if (lnx) hipLaunchKernelGGL((wgrad_group_kernel<float, false, true>), dim3(grid), dim3(768), 0, stream, a);
  else hipLaunchKernelGGL((wgrad_group_kernel<float, false, false>), dim3(grid), dim3(768), 0, stream, a);
  VITPE_CHECK_LAUNCH();
}
