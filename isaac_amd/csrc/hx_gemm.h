// hx_gemm.h -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, 64-wide waves.
//
// One kernel template covers the three dense products of an MLP layer (reference: nn.Linear + nn.ELU in
// humanoid/algo/ppo/actor_critic.py:57-80 and their autograd):
//   FWD    Y[M,N]  = elu(X[M,K] W[N,K]^T + b)         A K-major, B K-major,  epilogue bias+ELU
//   DGRAD  dX[M,N] = (dZ[M,K] W[K,N]) * elu'(H[M,N])  A K-major, B N-major,  epilogue * elu'(H)
//   WGRAD  dW[M,N] = dZ[K,M]^T X[K,N]   (split-K)     A M-major, B N-major,  epilogue -> partial slab
// (M,N,K are always the GEMM's own output rows / output cols / reduction length.)
//
// Tile: BM x BN x BK (BK = 16 or 32) per 256-thread workgroup (4 waves as 2x2), each wave (BM/2)x(BN/2) as 32x32 MFMA
// tiles.  fp32 MFMA is exact fp32 (k-ordered fma chain) and runs at the fp32 vector rate, so LDS and
// global bandwidth are never the limiter; the layout work here is about keeping every LDS access
// conflict-free and every global access a whole 64-byte segment:
//   K-major operand: global float4 along K -> ds_write_b128 into [row][16+4]; fragment = one ds_read_b128
//                    per lane holding k = 4h..4h+3 of an 8-deep block (h = lane>>5), feeding 4 MFMAs;
//   M/N-major operand: global float4 along the row index -> ds_write_b128 into [k][rows]; fragment =
//                    4 ds_read_b32 at k = 4h+j.
// Both use the same k <-> (lane half, MFMA j) assignment (MFMA j consumes k = j and k = 4+j of the block),
// which is legal because a dot product does not care in which order its k terms are visited.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_BIAS_ELU = 0, EPI_ELU_GRAD = 1, EPI_SLAB = 2, EPI_BIAS = 3 };

struct GemmArgs {
  const float* A; int lda;
  const float* B; int ldb;
  float* C; int ldc;
  int M, N, K;
  const float* bias;        // EPI_BIAS_ELU / EPI_BIAS: [N]
  const float* H; int ldh;  // EPI_ELU_GRAD: activations of the layer whose pre-activation gradient is produced
  int splits, kchunk;       // EPI_SLAB: reduction split; C is [splits][M][ldc]
  float* dbias;             // EPI_SLAB: [splits * db_parts][M] partial column sums of A (bias gradient), nullable
  int db_parts;             // EPI_SLAB: the column-sum work of a (tile_m, split) is shared by the first db_parts tile_n blocks
  int tiles_m, tiles_n;
  // POLL instantiations (the rollout's background critic): while *pause > 0 -- a foreground kernel that wants the CU's matrix
  // pipes to itself is running (the fused actor counts its workgroups in and out) -- the waves sleep between K tiles instead
  // of issuing.  The value is read one K tile ahead of its use (no wait on the critical path); the sleep is bounded.
  const int* pause;
  // Gathered operand rows (single-frame observation storage): when the kernel is instantiated with GA (A, K-major)
  // the operand is not a matrix in memory but a table of row starts -- row r begins at  base + off[r]  (a float offset, 4-byte
  // aligned only: 15 consecutive 41-wide frames of one robot) and its elements  [0, kz[r])  and  [klim, ...)  read as zero
  // (frames older than the robot's last reset; the padding columns).
  const int* a_off; const int* a_kz; int a_klim;
  // GA only: the assembled rows are also written out as an ordinary matrix [M][a_copy_ld] (16-byte aligned rows), so that the
  // weight-gradient product of the same layer reads them with the plain loader.  The tile_n workgroups of a row block share
  // the work (workgroup tile_n writes the K tiles kt = tile_n, tile_n + tiles_n, ...).  nullptr: no copy.
  float* a_copy; int a_copy_ld;
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a float4 that is only float-aligned in memory

#define HX_KPAD 4

// ELU(alpha=1).  exp(x)-1 through the hardware exp2 (abs error < 1.2e-7 for x <= 0, i.e. fp32 round-off of the
// activation's own scale); the libm expm1f it replaces cost ~10 % of a short-K GEMM's epilogue.
__device__ __forceinline__ float hx_elu(float x) { return x > 0.f ? x : (__expf(x) - 1.0f); }

#ifdef HX_GEMM_WAVES
#define HX_GEMM_OCC __attribute__((amdgpu_waves_per_eu(HX_GEMM_WAVES, HX_GEMM_WAVES)))
#else
#define HX_GEMM_OCC
#endif
// KFULL: the host guarantees that every reduction range is a whole number of K tiles (no partial-tile path in the loop)
template <int BM, int BN, int HX_BK, bool A_KM, bool B_KM> struct GemmLds {
  static constexpr int A_ELEMS = A_KM ? BM * (HX_BK + HX_KPAD) : HX_BK * BM;
  static constexpr int B_ELEMS = B_KM ? BN * (HX_BK + HX_KPAD) : HX_BK * BN;
  static constexpr int FLOATS = 2 * (A_ELEMS + B_ELEMS);
};

// One output tile (`logical` = tile index, times the split for EPI_SLAB), start to finish, by the whole workgroup.
__device__ __forceinline__ int hx_pause_load(const int* p) { return __builtin_nontemporal_load(p); }
// sleeps while the flag read one K tile ago is up; returns a fresh request for the next check
__device__ __forceinline__ void hx_pause_poll(const int* p, int& seen) {
  if (seen < 0) return;                            // gave up once: this workgroup never waits again
  int budget = 2048;                               // x (sleep 64 x 64 cycles + one L2 round trip) ~ 5 ms at most: never a hang
  while (seen > 0 && budget > 0) { --budget; __builtin_amdgcn_s_sleep(64); seen = *reinterpret_cast<const volatile int*>(p); }
  seen = (budget > 0) ? hx_pause_load(p) : -1;
}
// `paused_io` (POLL instantiations): the workgroup's pause state, kept by the persistent kernel ACROSS its tiles -- a workgroup that
// ran out of its sleep budget once (-1) never waits again, in this tile or any later one, so a lost decrement of the flag costs
// one bounded wait per workgroup and launch, not one per tile
template <int BM, int BN, int HX_BK, bool A_KM, bool B_KM, int EPI, bool KFULL, bool GA = false, bool GB = false, bool POLL = false>
__device__ __forceinline__ void hx_gemm_tile(const GemmArgs& g, const int logical, float* __restrict__ lds, int* paused_io = nullptr) {
  constexpr int WTM = BM / 2, WTN = BN / 2;       // per-wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;     // 32x32 MFMA tiles per wave
  constexpr int A_ELEMS = GemmLds<BM, BN, HX_BK, A_KM, B_KM>::A_ELEMS;
  constexpr int B_ELEMS = GemmLds<BM, BN, HX_BK, A_KM, B_KM>::B_ELEMS;
  constexpr int A_LOADS = BM * HX_BK / 4 / 256;   // float4 per thread per tile
  constexpr int B_LOADS = BN * HX_BK / 4 / 256;
  constexpr bool PIPE_HALVES = (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS);   // forward products only (see the main loop)
  const int tiles_mn = g.tiles_m * g.tiles_n;
  const int split = logical / tiles_mn;
  const int t = logical % tiles_mn;
  const int tile_m = t / g.tiles_n, tile_n = t % g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int k_begin = 0, k_end = g.K;
  if (EPI == EPI_SLAB) { k_begin = split * g.kchunk; k_end = min(g.K, k_begin + g.kchunk); }
  const int nk = (k_end - k_begin + HX_BK - 1) / HX_BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r32 = lane & 31;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[A_LOADS], rb[B_LOADS];
  // Bias gradient = column sums of dZ = row sums over k of the A tile.  Every tile_n block of a (tile_m, split) sees the
  // same A tiles; block tile_n takes K tiles tile_n, tile_n + P, ... (P = db_parts) and writes its own partial row, so no
  // block carries the whole side job (one owner made the launch 4 % longer: all workgroups end with the slowest).
  const bool db_owner = (EPI == EPI_SLAB) && !A_KM && (g.dbias != nullptr) && (tile_n < g.db_parts) && (tid < BM);
  int db_next = tile_n;
  float dbacc = 0.f;

  // Global -> register staging of one K tile.  The MFMA loop must stay nearly free of VALU work and of divergent
  // regions (tools/mfma_peak.py, profiles/r01_n_mfma_probe.txt: a probe with this loop's ingredients sustains 0.91 of the
  // matrix-pipe peak, 0.80 once its loads are guarded; per-tile 64-bit address arithmetic costs about as much).  So:
  //   * every lane owns fixed (row, k-offset) slots; their addresses are computed ONCE and advanced by a constant stride;
  //   * rows / columns past the matrix edge alias a valid row (their products land in output elements that the epilogue
  //     never stores), so no lane is ever masked off;
  //   * only the last tile of a reduction can be partial in k: a uniform branch takes the zero-filling path there.
  static_assert(!GA || A_KM, "gathered A rows are K-major (forward products)");
  static_assert(!GB, "gathered B rows were measured 14 % slower than the plain loader and are gone: the forward product writes the assembled rows out instead (a_copy)");
  const float* pa[A_LOADS]; const float* pb[B_LOADS];
  int ka[A_LOADS], kb_[B_LOADS];                    // k offset of the slot inside a tile
  int za[A_LOADS]; unsigned wa[A_LOADS];            // GA: first valid k of the slot's row and the number of valid k from there (klim - kz)
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int idx = tid + i * 256;
    if (A_KM) {
      const int row = idx / (HX_BK / 4), k4 = idx % (HX_BK / 4);
      ka[i] = k4 * 4;
      const int gr = min(m0 + row, g.M - 1);
      if (GA) { pa[i] = g.A + g.a_off[gr] + k_begin + k4 * 4; za[i] = g.a_kz[gr]; wa[i] = (unsigned)(g.a_klim - za[i]); }
      else pa[i] = g.A + (size_t)gr * g.lda + k_begin + k4 * 4;
    } else {
      const int k = idx / (BM / 4), m4 = idx % (BM / 4);
      const int gm = m0 + m4 * 4;
      ka[i] = k;
      pa[i] = g.A + (size_t)(k_begin + k) * g.lda + (gm < g.M ? gm : 0);
    }
  }
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int idx = tid + i * 256;
    if (B_KM) {
      const int row = idx / (HX_BK / 4), k4 = idx % (HX_BK / 4);
      kb_[i] = k4 * 4;
      pb[i] = g.B + (size_t)min(n0 + row, g.N - 1) * g.ldb + k_begin + k4 * 4;
    } else {
      const int k = idx / (BN / 4), n4 = idx % (BN / 4);
      const int gn = n0 + n4 * 4;
      kb_[i] = k;
      pb[i] = g.B + (size_t)(k_begin + k) * g.ldb + (gn < g.N ? gn : 0);
    }
  }
  unsigned mka[A_LOADS];      // GA: mask parameter of the tile that is in the staging registers
  int lt = 0;                 // K tile index of the staging registers
  const size_t stride_a = A_KM ? (size_t)HX_BK : (size_t)HX_BK * g.lda;
  const size_t stride_b = B_KM ? (size_t)HX_BK : (size_t)HX_BK * g.ldb;
  auto load_tile = [&](int kt) {        // tiles must be requested in order 0, 1, 2, ...: the slot pointers advance
    const int k0 = k_begin + kt * HX_BK;
    if (GA) {
      // Gathered rows.  The loads are issued here and NOT touched: the zero prefix / padding mask is applied when the tile is
      // written to LDS (store_tile), one iteration later -- masking at load time makes the wave wait for the data it has just
      // requested and exposes the whole global latency in every K tile (measured: +9-14 % on the update's products, +45 % on
      // the rollout critic's one-wave-per-SIMD grid, profiles/r03_c).  What the mask needs is remembered per slot (mka).
      const bool whole = KFULL || k0 + HX_BK <= k_end;
      lt = kt;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        ra[i] = *reinterpret_cast<const f32x4u*>(pa[i]); pa[i] += stride_a;
        mka[i] = (unsigned)(k0 - k_begin + ka[i] - za[i]);          // k of element 0 relative to the first valid one; valid: 0 <= . < wa
      }
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i) {
        if (whole) { rb[i] = *reinterpret_cast<const f32x4*>(pb[i]); pb[i] += stride_b; }
        else { f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (k0 + kb_[i] < k_end) v = *reinterpret_cast<const f32x4*>(pb[i]); rb[i] = v; }
      }
    } else if (KFULL || k0 + HX_BK <= k_end) { // uniform: whole tile inside the reduction range
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) { ra[i] = *reinterpret_cast<const f32x4*>(pa[i]); pa[i] += stride_a; }
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i) { rb[i] = *reinterpret_cast<const f32x4*>(pb[i]); pb[i] += stride_b; }
    } else {                            // the one partial tile at the end: k positions past k_end contribute zeros
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + ka[i] < k_end) v = *reinterpret_cast<const f32x4*>(pa[i]);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + kb_[i] < k_end) v = *reinterpret_cast<const f32x4*>(pb[i]);
        rb[i] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* As = lds + buf * (A_ELEMS + B_ELEMS);
    float* Bs = As + A_ELEMS;
    if (GA) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[i][j] = (mka[i] + (unsigned)j < wa[i]) ? ra[i][j] : 0.f;
      if (g.a_copy != nullptr && (lt % g.tiles_n) == tile_n) {       // uniform: this workgroup's share of the row block's K tiles
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
          const int idx = tid + i * 256, row = idx / (HX_BK / 4), k = lt * HX_BK + ka[i];
          if (m0 + row < g.M && k < g.a_copy_ld) *reinterpret_cast<f32x4*>(g.a_copy + (size_t)(m0 + row) * g.a_copy_ld + k) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int idx = tid + i * 256;
      if (A_KM) { const int row = idx / (HX_BK / 4), k4 = idx % (HX_BK / 4); *reinterpret_cast<f32x4*>(As + row * (HX_BK + HX_KPAD) + k4 * 4) = ra[i]; }
      else { const int k = idx / (BM / 4), m4 = idx % (BM / 4); *reinterpret_cast<f32x4*>(As + k * BM + m4 * 4) = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const int idx = tid + i * 256;
      if (B_KM) { const int row = idx / (HX_BK / 4), k4 = idx % (HX_BK / 4); *reinterpret_cast<f32x4*>(Bs + row * (HX_BK + HX_KPAD) + k4 * 4) = rb[i]; }
      else { const int k = idx / (BN / 4), n4 = idx % (BN / 4); *reinterpret_cast<f32x4*>(Bs + k * BN + n4 * 4) = rb[i]; }
    }
  };
  // fragments of one HALF of a K tile (HX_BK/2 deep = KB8 blocks of 8): A and B operands of 4 MFMA steps per block
  constexpr int KB8 = HX_BK / 16;
  struct Frags { f32x4 a[KB8][TM], b[KB8][TN]; };
  auto read_frags = [&](int buf, int half, Frags& f) {
    const float* As = lds + buf * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int q = 0; q < KB8; ++q) {
      const int kb = half * KB8 + q;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + r32;
        if (A_KM) f.a[q][i] = *reinterpret_cast<const f32x4*>(As + row * (HX_BK + HX_KPAD) + kb * 8 + 4 * h);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) f.a[q][i][j] = As[(kb * 8 + 4 * h + j) * BM + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WTN + i * 32 + r32;
        if (B_KM) f.b[q][i] = *reinterpret_cast<const f32x4*>(Bs + row * (HX_BK + HX_KPAD) + kb * 8 + 4 * h);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) f.b[q][i][j] = Bs[(kb * 8 + 4 * h + j) * BN + row];
        }
      }
    }
  };
  auto mfma_half = [&](const Frags& f) {
#pragma unroll
    for (int q = 0; q < KB8; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q][a][j], f.b[q][b][j], acc[a][b], 0, 0, 0);
  };
  auto bias_grad = [&](int buf, int kt) {
    if (EPI == EPI_SLAB && !A_KM) {
      if (kt == db_next) {                                   // uniform
        db_next += g.db_parts;
        if (db_owner) {
          const float* As = lds + buf * (A_ELEMS + B_ELEMS);
          float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;      // four chains instead of one 16-deep dependent chain
#pragma unroll
          for (int k = 0; k < HX_BK; k += 4) { p0 += As[k * BM + tid]; p1 += As[(k + 1) * BM + tid]; p2 += As[(k + 2) * BM + tid]; p3 += As[(k + 3) * BM + tid]; }
          dbacc += (p0 + p1) + (p2 + p3);
        }
      }
    }
  };

  // Main loop, software-pipelined over HALF tiles so that the matrix pipe always has register-resident operands on
  // both sides of the one barrier per K tile:
  //   top:    ds_read  second-half fragments of tile kt            (latency hidden by the first-half MFMAs)
  //           MFMA     first half of tile kt                       (fragments read before the previous barrier)
  //           ds_write tile kt+1 (staged in registers one iteration ago) into the other LDS buffer
  //           global   loads of tile kt+2 -> staging registers     (a whole iteration to arrive)
  //           barrier
  //           ds_read  first-half fragments of tile kt+1           (latency hidden by the second-half MFMAs)
  //           MFMA     second half of tile kt
  // The buffer written in iteration kt was last read before the barrier of iteration kt-1, so one barrier suffices.
  if (PIPE_HALVES && nk > 0) {
    Frags f0, f1;
    load_tile(0);
    store_tile(0);
    if (nk > 1) load_tile(1);
    __syncthreads();
    read_frags(0, 0, f0);
    int paused = (POLL && paused_io) ? *paused_io : 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (POLL && g.pause != nullptr) hx_pause_poll(g.pause, paused);
      read_frags(cur, 1, f1);
      bias_grad(cur, kt);
      mfma_half(f0);
      if (kt + 1 < nk) store_tile(cur ^ 1);
      if (kt + 2 < nk) load_tile(kt + 2);
      __syncthreads();
      if (kt + 1 < nk) read_frags(cur ^ 1, 0, f0);
      mfma_half(f1);
    }
    if (POLL && paused_io) *paused_io = paused;
  }

  auto compute = [&](int buf, int kt) {
    const float* As = lds + buf * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int kb = 0; kb < HX_BK / 8; ++kb) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + r32;
        if (A_KM) fa[i] = *reinterpret_cast<const f32x4*>(As + row * (HX_BK + HX_KPAD) + kb * 8 + 4 * h);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[i][j] = As[(kb * 8 + 4 * h + j) * BM + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WTN + i * 32 + r32;
        if (B_KM) fb[i] = *reinterpret_cast<const f32x4*>(Bs + row * (HX_BK + HX_KPAD) + kb * 8 + 4 * h);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[i][j] = Bs[(kb * 8 + 4 * h + j) * BN + row];
        }
      }
#ifdef HX_GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
#ifdef HX_GEMM_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    bias_grad(buf, kt);
  };

  // Plain double-buffered loop (two barriers' worth of exposed LDS latency per K tile, but fewer live registers):
  // measured faster for the backward products (tools/gemm_bench.py, profiles/r01_e_gemm_loops.txt).
  if (!PIPE_HALVES && nk > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int paused = (POLL && paused_io) ? *paused_io : 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1 < nk);
      if (POLL && g.pause != nullptr) hx_pause_poll(g.pause, paused);
      if (more) load_tile(kt + 1);
      compute(kt & 1, kt);
      if (more) store_tile((kt + 1) & 1);
      __syncthreads();
    }
    if (POLL && paused_io) *paused_io = paused;
  }

  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  // Interior tiles take a branch-free path: with per-element guards hipcc puts an `s_waitcnt vmcnt(0)` in front of
  // every guarded block, and because vmcnt counts stores on CDNA4 that serialises all 64 stores of a lane
  // (measured: ~20-35 % of a tile's time).  Straight-line code lets loads batch and stores stream.
  float* Cb = g.C;
  if (EPI == EPI_SLAB) Cb += (size_t)split * g.M * g.ldc;
  const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  if (interior) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * WTN + b * 32 + r32;
        const int row0 = m0 + wm * WTM + a * 32 + 4 * h;
        float bv = 0.f;
        if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) bv = g.bias[col];
        float hv[16];
        if (EPI == EPI_ELU_GRAD) {
#pragma unroll
          for (int e = 0; e < 16; ++e) hv[e] = g.H[(size_t)(row0 + (e & 3) + 8 * (e >> 2)) * g.ldh + col];
        }
        float out[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[a][b][e];
          if (EPI == EPI_BIAS_ELU) v = hx_elu(v + bv);
          if (EPI == EPI_BIAS) v = v + bv;
          if (EPI == EPI_ELU_GRAD) v = v * (hv[e] > 0.f ? 1.f : hv[e] + 1.f);
          out[e] = v;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) Cb[(size_t)(row0 + (e & 3) + 8 * (e >> 2)) * g.ldc + col] = out[e];
      }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * WTN + b * 32 + r32;
        if (col >= g.N) continue;
        float bv = 0.f;
        if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) bv = g.bias[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * WTM + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row >= g.M) continue;
          float v = acc[a][b][e];
          if (EPI == EPI_BIAS_ELU) v = hx_elu(v + bv);
          if (EPI == EPI_BIAS) v = v + bv;
          if (EPI == EPI_ELU_GRAD) {
            const float hh = g.H[(size_t)row * g.ldh + col];
            v = v * (hh > 0.f ? 1.f : hh + 1.f);
          }
          Cb[(size_t)row * g.ldc + col] = v;
        }
      }
  }
  if (EPI == EPI_SLAB && !A_KM) {
    if (db_owner && m0 + tid < g.M) g.dbias[((size_t)split * g.db_parts + tile_n) * g.M + m0 + tid] = dbacc;
  }
}

template <int BM, int BN, int HX_BK, bool A_KM, bool B_KM, int EPI, bool KFULL = false, bool GA = false, bool GB = false>
__global__ void __launch_bounds__(256) HX_GEMM_OCC hx_gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[GemmLds<BM, BN, HX_BK, A_KM, B_KM>::FLOATS];
  // XCD-aware block -> tile map: blocks b, b+8, b+16, ... share an XCD (and its L2); give each XCD a
  // contiguous run of logical tiles so neighbours re-use the same A rows out of L2.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  int logical;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  hx_gemm_tile<BM, BN, HX_BK, A_KM, B_KM, EPI, KFULL, GA, GB>(g, logical, lds);
}

// Several products in ONE launch.
// (1) The weight gradients of a minibatch (split-K).  Launched layer by layer, every product gets its own full wave of
// workgroups: the 256 x 128 layer (2 tiles) is cut into 240 slices of 256 rows and every layer leaves ~50 MB of partial slabs
// whatever its size.  In a group all members are cut into the SAME number of slices, chosen so that the tiles of all members
// together fill the chip once (hector: five layers, 44 tiles x 17 slices): every workgroup runs the same long K loop, the
// slabs shrink with the slice count, and four launch boundaries go.
// (2) The same layer of the actor and of the critic (forward and input-gradient products): two short launches of 1.25 rounds
// of workgroups each become one of 2.5, with one ramp and one tail.
// `first[i]` = first logical block of member i; a member's blocks are numbered like a launch of its own (hx_gemm_tile's
// `logical`: split-major for split-K).  Block -> work map: blocks b, b + 8, ... share an XCD and its L2; XCD x walks its
// contiguous share of member 0's tiles, then its share of member 1's, ... -- every XCD gets the same mix of long and short
// tiles, and neighbours in an XCD's queue re-use the same A rows.  Grid = 8 x the longest queue; the few surplus blocks exit.
#define HX_GROUP_MAX 8
struct GemmGroup { GemmArgs p[HX_GROUP_MAX]; int first[HX_GROUP_MAX + 1]; int n; };
// Block -> work map of a grouped launch.  Member m has T_m blocks; XCD x (blocks x, x + 8, ...: they share an L2) takes a
// contiguous run of  T_m / 8  of them, and the  T_m % 8  left over go one each to the XCDs  start_m, start_m + 1, ...  where
// start_m continues where the previous member's leftovers ended -- so the XCDs' totals differ by at most one block.  (With
// the leftovers always on XCDs 0, 1, ... a launch sized for one workgroup per CU put 34 workgroups on the first XCD's 32 CUs
// and ran twice as long, profiles/r04_b_gemm_lab.txt.)
static inline __host__ __device__ int hx_group_count(int T, int start, int xcd) { return (T >> 3) + ((((xcd - start) & 7) < (T & 7)) ? 1 : 0); }
static inline int hx_group_grid(const GemmGroup& G) {
  int longest = 0;
  for (int x = 0; x < 8; ++x) {
    int len = 0, start = 0;
    for (int m = 0; m < G.n; ++m) { const int T = G.first[m + 1] - G.first[m]; len += hx_group_count(T, start, x); start = (start + T) & 7; }
    if (len > longest) longest = len;
  }
  return 8 * longest;
}
// member `pi` (-1: surplus block) and its logical block for blockIdx.x
__device__ __forceinline__ void hx_group_pick(const GemmGroup& G, int& pi, int& logical) {
  const int xcd = blockIdx.x & 7;
  int idx = blockIdx.x >> 3, start = 0;
  pi = -1; logical = 0;
#pragma unroll
  for (int m = 0; m < HX_GROUP_MAX; ++m) {
    if (m < G.n && pi < 0) {
      const int T = G.first[m + 1] - G.first[m], q = T >> 3, r = T & 7, xr = (xcd - start) & 7;
      const int cnt = q + (xr < r ? 1 : 0);
      if (idx < cnt) { pi = m; logical = xr * q + min(xr, r) + idx; }
      else idx -= cnt;
      start = (start + T) & 7;
    }
  }
}
template <int BM, int BN, int HX_BK, bool A_KM, bool B_KM, int EPI, bool KFULL = false>
__global__ void __launch_bounds__(256) HX_GEMM_OCC hx_gemm_group_kernel(GemmGroup G) {
  __shared__ __attribute__((aligned(16))) float lds[GemmLds<BM, BN, HX_BK, A_KM, B_KM>::FLOATS];
  int pi, logical;
  hx_group_pick(G, pi, logical);
  if (pi < 0) return;
  // member of this block: a chain of uniform selects (indexing the kernel-argument array with a run-time value would move
  // the whole struct to scratch)
  GemmArgs g = G.p[0];
#pragma unroll
  for (int i = 1; i < HX_GROUP_MAX; ++i) if (pi == i) g = G.p[i];
  hx_gemm_tile<BM, BN, HX_BK, A_KM, B_KM, EPI, KFULL>(g, logical, lds);
}

// The same product with a SMALL fixed grid whose workgroups walk the tiles (tile t, t + grid, ...).  Not faster per se
// (profiles/README.md "persistent tiles"); its use is the background critic of the rollout: 128 workgroups settle on 128
// CUs, one wave per SIMD there, and leave the other CUs entirely free -- an env-step wave needs a whole SIMD's registers,
// and with an ordinary launch every SIMD of the chip soon holds a GEMM wave.
template <int BM, int BN, int HX_BK, bool A_KM, bool B_KM, int EPI, bool KFULL = false, bool GA = false, bool GB = false>
__global__ void __launch_bounds__(256) hx_gemm_persistent_kernel(GemmArgs g, int total_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[GemmLds<BM, BN, HX_BK, A_KM, B_KM>::FLOATS];
  int paused = 0;
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    hx_gemm_tile<BM, BN, HX_BK, A_KM, B_KM, EPI, KFULL, GA, GB, true>(g, t, lds, &paused);
    __syncthreads();          // the next tile's first LDS stores must not overtake this tile's last fragment reads
  }
}
