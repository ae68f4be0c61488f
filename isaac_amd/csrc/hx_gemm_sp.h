// hx_gemm_sp.h -- the products of hx_gemm.h with a hand-placed main loop ("sp" = slot-placed).
//
// Same arithmetic as hx_gemm_tile (same MFMA, same k order inside an 8-deep block, same tile shapes): results are bit for bit
// those of hx_gemm.h.  What differs is WHERE the non-matrix instructions of a K tile sit.  v_mfma_f32_32x32x2_f32 holds the
// SIMD's matrix pipe for 64 cycles and a wave issues in order, so after a wave's last MFMA of a run the pipe has 64 cycles of
// work left: whatever the wave does before its next MFMA -- fragment reads and the s_waitcnt behind them, the LDS stores of
// the next tile, the workgroup barrier -- is a bubble as soon as it takes longer than that, and hipcc schedules all of it as
// clusters between groups of four MFMAs (profiles/r04_a_gemm_loop.txt).  Here every MFMA is a *slot* that carries at most one
// LDS read, one LDS store and one global load, pinned with sched_barrier(0):
//   * fragments are double-buffered per 8-deep k block: the reads of block b+1 ride in the first slots of block b, so no
//     MFMA waits for an LDS round trip it has just asked for;
//   * the next tile's LDS stores and the global loads of the tile after it ride in the first slots of a tile (one per slot);
//   * the one barrier per K tile sits a few slots into the tile's LAST block -- its own stores were issued a block or more
//     earlier, so it waits for skew only -- and the first fragments of the next tile are read behind it, in front of the
//     block's remaining MFMAs.
// Operand kinds as in hx_gemm.h (K-major: [row][BK+4], one ds_read_b128 per 32x32 tile and block; row-major: [k][rows]).  A
// row-major operand maps the 32 rows of MFMA tile i of a wave to rows  T*r + i  (T = tiles per wave in that direction) instead
// of  32*i + r : a lane's T operands of one k are then contiguous in LDS (one ds_read_b64 for T = 2) and its T outputs of one
// row contiguous in memory (one 8-byte store).  Which rows a tile covers changes nothing in any sum.
#pragma once
#include <type_traits>
#include "hx_gemm.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int WM, int WN> struct GemmSp {
  static constexpr int NT = 64 * WM * WN;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static constexpr int A_ELEMS = A_KM ? BM * (BKT + HX_KPAD) : BKT * BM;
  static constexpr int B_ELEMS = B_KM ? BN * (BKT + HX_KPAD) : BKT * BN;
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr int FLOATS = 2 * STAGE;
  static constexpr int A_F4 = BM * BKT / 4, B_F4 = BN * BKT / 4;          // float4 slots of a K tile
  static constexpr int A_LOADS = (A_F4 + NT - 1) / NT, B_LOADS = (B_F4 + NT - 1) / NT;
  static constexpr bool A_PART = (A_F4 % NT) != 0, B_PART = (B_F4 % NT) != 0;   // the last slot exists for the first waves only (narrow strips)
  static constexpr int NW = A_LOADS + B_LOADS;
  static constexpr int NB = BKT / 8;                   // 8-deep k blocks per tile
  static constexpr int MF = 4 * TM * TN;               // MFMAs (slots) per block
  static constexpr int RA = A_KM ? TM : 4, RB = B_KM ? TN : 4, R = RA + RB;    // LDS fragment reads per block
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiles are whole 32x32 MFMA tiles");
  static_assert((A_F4 % 64) == 0 && (B_F4 % 64) == 0, "a wave is inside or outside a staging slot as a whole");
  static_assert(NB % 2 == 0, "fragment buffers alternate per block");
  static_assert(A_KM || TM == 1 || TM == 2 || TM == 4, "row-major fragment read widths");
  static_assert(B_KM || TN == 1 || TN == 2 || TN == 4, "row-major fragment read widths");
  static_assert(2 * NW <= NB * MF && R + 4 <= MF, "a tile has a slot for every staging operation, a block for every fragment read");
};

// iteration kinds of the K loop (the steady state carries no conditional code)
enum { SP_STEADY = 0, SP_LASTLOAD = 1, SP_WRITEONLY = 2, SP_FINAL = 3 };

// what a call does with its accumulators (stream-K segments, hx_gemm_sk_kernel): SP_WHOLE the product's epilogue; SP_PART_OUT no
// epilogue, the raw accumulators go to `partial` ([register][thread], coalesced); SP_PART_IN `partial` (another workgroup's part of the
// same tile, same layout) is added to them before the epilogue
enum { SP_WHOLE = 0, SP_PART_OUT = 1, SP_PART_IN = 2 };

// One tile over the reduction range [k_begin, k_end) (k_begin a multiple of the K tile; only the reduction's last K tile may be partial).
// POLL (the rollout's background critic): while *g.pause > 0 the workgroup sleeps between K tiles (hx_pause_poll, hx_gemm.h); the
// gave-up state lives in *paused_io across the tiles of a persistent workgroup.
template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL, bool POLL = false>
__device__ __forceinline__ void hx_gemm_tile_sp_ex(const GemmArgs& g, const int tile_m, const int tile_n, const int split, const int k_begin, const int k_end,
                                                   float* __restrict__ lds, const int mode = SP_WHOLE, float* __restrict__ partial = nullptr, int* paused_io = nullptr) {
  using P = GemmSp<BM, BN, BKT, A_KM, B_KM, WM, WN>;
  constexpr int NT = P::NT, WTM = P::WTM, WTN = P::WTN, TM = P::TM, TN = P::TN;
  constexpr int A_ELEMS = P::A_ELEMS, STAGE = P::STAGE, A_LOADS = P::A_LOADS, B_LOADS = P::B_LOADS, NW = P::NW;
  constexpr int NB = P::NB, MF = P::MF, RA = P::RA, R = P::R;
  constexpr int BARRIER_SLOT = 2;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (k_end - k_begin + BKT - 1) / BKT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, r32 = lane & 31;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // bias gradient = column sums of the A tiles (hx_gemm.h): thread t owns tile rows t, t + NT, ...
  constexpr int DBN = (BM + NT - 1) / NT;
  const bool db_owner = (EPI == EPI_SLAB) && !A_KM && (g.dbias != nullptr) && (tile_n < g.db_parts);
  int db_next = tile_n;
  float dbacc[DBN];
#pragma unroll
  for (int c = 0; c < DBN; ++c) dbacc[c] = 0.f;

  // ---- global -> register staging: fixed (row, k) slots per lane, addresses advanced by a constant stride (hx_gemm.h)
  f32x4 ra[A_LOADS], rb[B_LOADS];
  const float* pa[A_LOADS]; const float* pb[B_LOADS];
  int ka[A_LOADS], kb_[B_LOADS];
  int wa_off[A_LOADS], wb_off[B_LOADS];            // LDS float offset of the slot inside a stage
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int idx = tid + i * NT;
    if (A_KM) {
      const int row = idx / (BKT / 4), k4 = idx % (BKT / 4);
      ka[i] = k4 * 4;
      pa[i] = g.A + (size_t)min(m0 + row, g.M - 1) * g.lda + k_begin + k4 * 4;
      wa_off[i] = row * (BKT + HX_KPAD) + k4 * 4;
    } else {
      const int k = idx / (BM / 4), m4 = idx % (BM / 4);
      const int gm = m0 + m4 * 4;
      ka[i] = k;
      pa[i] = g.A + (size_t)(k_begin + k) * g.lda + (gm < g.M ? gm : 0);
      wa_off[i] = k * BM + m4 * 4;
    }
  }
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int idx = tid + i * NT;
    if (B_KM) {
      const int row = idx / (BKT / 4), k4 = idx % (BKT / 4);
      kb_[i] = k4 * 4;
      pb[i] = g.B + (size_t)min(n0 + row, g.N - 1) * g.ldb + k_begin + k4 * 4;
      wb_off[i] = A_ELEMS + row * (BKT + HX_KPAD) + k4 * 4;
    } else {
      const int k = idx / (BN / 4), n4 = idx % (BN / 4);
      const int gn = n0 + n4 * 4;
      kb_[i] = k;
      pb[i] = g.B + (size_t)(k_begin + k) * g.ldb + (gn < g.N ? gn : 0);
      wb_off[i] = A_ELEMS + k * BN + n4 * 4;
    }
  }
  const size_t stride_a = A_KM ? (size_t)BKT : (size_t)BKT * g.lda;
  const size_t stride_b = B_KM ? (size_t)BKT : (size_t)BKT * g.ldb;
  // one staging load (idx < A_LOADS: operand A); `guard`: this is the reduction's last, possibly partial tile
  auto load_one = [&](int idx, int kt, bool guard) {
    const int k0 = k_begin + kt * BKT;
    if (idx < A_LOADS) {
      const int i = idx;
      if (P::A_PART && i == A_LOADS - 1 && tid + i * NT >= P::A_F4) return;
      if (!KFULL && guard) { f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (k0 + ka[i] < k_end) v = *reinterpret_cast<const f32x4*>(pa[i]); ra[i] = v; }
      else { ra[i] = *reinterpret_cast<const f32x4*>(pa[i]); pa[i] += stride_a; }
    } else {
      const int i = idx - A_LOADS;
      if (P::B_PART && i == B_LOADS - 1 && tid + i * NT >= P::B_F4) return;
      if (!KFULL && guard) { f32x4 v = {0.f, 0.f, 0.f, 0.f}; if (k0 + kb_[i] < k_end) v = *reinterpret_cast<const f32x4*>(pb[i]); rb[i] = v; }
      else { rb[i] = *reinterpret_cast<const f32x4*>(pb[i]); pb[i] += stride_b; }
    }
  };
  auto write_one = [&](int idx, int buf) {
    float* S = lds + buf * STAGE;
    if (idx < A_LOADS) { if (P::A_PART && idx == A_LOADS - 1 && tid + idx * NT >= P::A_F4) return; *reinterpret_cast<f32x4*>(S + wa_off[idx]) = ra[idx]; }
    else { const int i = idx - A_LOADS; if (P::B_PART && i == B_LOADS - 1 && tid + i * NT >= P::B_F4) return; *reinterpret_cast<f32x4*>(S + wb_off[i]) = rb[i]; }
  };

  // ---- fragments: fa[p][i][j] = A operand of MFMA step j (k = 8 kb + j in lane half 0, + 4 in half 1) for tile i
  float fa[2][TM][4], fb[2][TN][4];
  const int fa_base = A_KM ? (wm * WTM + r32) * (BKT + HX_KPAD) + 4 * h : (4 * h) * BM + wm * WTM + TM * r32;
  const int fb_base = A_ELEMS + (B_KM ? (wn * WTN + r32) * (BKT + HX_KPAD) + 4 * h : (4 * h) * BN + wn * WTN + TN * r32);
  auto read_one = [&](int p, int buf, int kb, int idx) {
    const float* S = lds + buf * STAGE;
    if (idx < RA) {
      if (A_KM) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + fa_base + idx * 32 * (BKT + HX_KPAD) + kb * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[p][idx][j] = v[j];
      } else {
        const float* q = S + fa_base + (kb * 8 + idx) * BM;
        if (TM == 1) fa[p][0][idx] = q[0];
        else if (TM == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(q); fa[p][0][idx] = v[0]; fa[p][1][idx] = v[1]; }
        else { const f32x4 v = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[p][i % TM][idx] = v[i]; }
      }
    } else {
      const int ib = idx - RA;
      if (B_KM) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(S + fb_base + ib * 32 * (BKT + HX_KPAD) + kb * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[p][ib][j] = v[j];
      } else {
        const float* q = S + fb_base + (kb * 8 + ib) * BN;
        if (TN == 1) fb[p][0][ib] = q[0];
        else if (TN == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(q); fb[p][0][ib] = v[0]; fb[p][1][ib] = v[1]; }
        else { const f32x4 v = *reinterpret_cast<const f32x4*>(q);
#pragma unroll
          for (int i = 0; i < 4; ++i) fb[p][i % TN][ib] = v[i]; }
      }
    }
  };
  auto bias_grad = [&](int buf, int kt) {
    if (EPI == EPI_SLAB && !A_KM) {
      if (kt == db_next) {                                   // uniform
        db_next += g.db_parts;
        if (db_owner) {
          const float* As = lds + buf * STAGE;
#pragma unroll
          for (int c = 0; c < DBN; ++c) {
            const int col = tid + c * NT;
            if (col < BM) {
              float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
              for (int k = 0; k < BKT; k += 4) { p0 += As[k * BM + col]; p1 += As[(k + 1) * BM + col]; p2 += As[(k + 2) * BM + col]; p3 += As[(k + 3) * BM + col]; }
              dbacc[c] += (p0 + p1) + (p2 + p3);
            }
          }
        }
      }
    }
  };

  // One K tile.  On entry: fragments of (kt, block 0) are in fa[0] / fb[0] (read behind the previous barrier); ra / rb hold
  // tile kt+1 (loads in flight).  KIND says what exists beyond this tile.
  int paused = (POLL && paused_io) ? *paused_io : 0;
  auto tile_iter = [&](auto kind_c, int kt) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool WRITE = (KIND != SP_FINAL), LOAD = (KIND == SP_STEADY || KIND == SP_LASTLOAD);
    const int cur = kt & 1;
    if (POLL && g.pause != nullptr) hx_pause_poll(g.pause, paused);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int p = b & 1;
#pragma unroll
      for (int s = 0; s < MF; ++s) {
        const int j = s / (TM * TN), a = (s / TN) % TM, bb = s % TN;
        acc[a][bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p][a][j], fb[p][bb][j], acc[a][bb], 0, 0, 0);
        if (b + 1 < NB) { if (s < R) read_one(p ^ 1, cur, b + 1, s); }                       // next block of this tile
        else if (WRITE && s > BARRIER_SLOT && s <= BARRIER_SLOT + R) read_one(p ^ 1, cur ^ 1, 0, s - BARRIER_SLOT - 1);   // first block of the next tile
        const int gs = b * MF + s;          // slot index inside the tile: stores first, then the loads that refill their registers
        if (WRITE && gs < NW) write_one(gs, cur ^ 1);
        if (LOAD && gs >= NW && gs < 2 * NW) load_one(gs - NW, kt + 2, KIND == SP_LASTLOAD);
        if (WRITE && b == NB - 1 && s == BARRIER_SLOT) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (b == 0) bias_grad(cur, kt);
    }
  };

  if (nk > 0) {
    // prologue: tile 0 -> LDS, tile 1 -> registers, first fragments
#pragma unroll
    for (int i = 0; i < NW; ++i) load_one(i, 0, nk == 1);
#pragma unroll
    for (int i = 0; i < NW; ++i) write_one(i, 0);
    if (nk > 1) {
#pragma unroll
      for (int i = 0; i < NW; ++i) load_one(i, 1, nk == 2);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i) read_one(0, 0, 0, i);
    int kt = 0;
    for (; kt + 3 < nk; ++kt) tile_iter(std::integral_constant<int, SP_STEADY>{}, kt);
    if (kt + 3 == nk) { tile_iter(std::integral_constant<int, SP_LASTLOAD>{}, kt); ++kt; }
    if (kt + 2 == nk) { tile_iter(std::integral_constant<int, SP_WRITEONLY>{}, kt); ++kt; }
    tile_iter(std::integral_constant<int, SP_FINAL>{}, kt);
  }
  if (POLL && paused_io) *paused_io = paused;

  if (mode == SP_PART_OUT) {             // uniform
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) partial[(size_t)((i * TN + j) * 16 + e) * NT + tid] = acc[i][j][e];
    return;
  }
  if (mode == SP_PART_IN) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += partial[(size_t)((i * TN + j) * 16 + e) * NT + tid];
  }
  // ---- epilogue.  C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* Cb = g.C;
  if (EPI == EPI_SLAB) Cb += (size_t)split * g.M * g.ldc;
  const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  auto out_row = [&](int a, int e) { const int mp = (e & 3) + 8 * (e >> 2) + 4 * h; return m0 + wm * WTM + (A_KM ? a * 32 + mp : TM * mp + a); };
  auto apply = [&](float v, float bv, float hv) {
    if (EPI == EPI_BIAS_ELU) v = hx_elu(v + bv);
    if (EPI == EPI_BIAS) v = v + bv;
    if (EPI == EPI_ELU_GRAD) v = v * (hv > 0.f ? 1.f : hv + 1.f);
    return v;
  };
  if (B_KM) {                      // a lane's columns: n0 + wn*WTN + 32 b + r32, one at a time
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * WTN + b * 32 + r32;
        if (interior) {
          float bv = 0.f;
          if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) bv = g.bias[col];
          float hv[16];
          if (EPI == EPI_ELU_GRAD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) hv[e] = g.H[(size_t)out_row(a, e) * g.ldh + col];
          }
          float out[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) out[e] = apply(acc[a][b][e], bv, EPI == EPI_ELU_GRAD ? hv[e] : 0.f);
#pragma unroll
          for (int e = 0; e < 16; ++e) Cb[(size_t)out_row(a, e) * g.ldc + col] = out[e];
        } else if (col < g.N) {
          float bv = 0.f;
          if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) bv = g.bias[col];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = out_row(a, e);
            if (row >= g.M) continue;
            const float hv = (EPI == EPI_ELU_GRAD) ? g.H[(size_t)row * g.ldh + col] : 0.f;
            Cb[(size_t)row * g.ldc + col] = apply(acc[a][b][e], bv, hv);
          }
        }
      }
  } else {                         // a lane's columns: n0 + wn*WTN + TN r32 + (0 .. TN-1), contiguous -> vector accesses
    typedef float vecT __attribute__((ext_vector_type(TN == 1 ? 1 : TN)));
    const int col0 = n0 + wn * WTN + TN * r32;
    vecT bv;
#pragma unroll
    for (int b = 0; b < TN; ++b) bv[b] = 0.f;
    if ((EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) && col0 < g.N) bv = *reinterpret_cast<const vecT*>(g.bias + col0);
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if (interior) {
        vecT hv[16];
        if (EPI == EPI_ELU_GRAD) {
#pragma unroll
          for (int e = 0; e < 16; ++e) hv[e] = *reinterpret_cast<const vecT*>(g.H + (size_t)out_row(a, e) * g.ldh + col0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          vecT o;
#pragma unroll
          for (int b = 0; b < TN; ++b) o[b] = apply(acc[a][b][e], bv[b], EPI == EPI_ELU_GRAD ? hv[e][b] : 0.f);
          *reinterpret_cast<vecT*>(Cb + (size_t)out_row(a, e) * g.ldc + col0) = o;      // (non-temporal slab stores: no difference, profiles/r04_b_gemm_lab.txt)
        }
      } else if (col0 < g.N) {     // N is a multiple of 4 and TN divides 4: a vector is inside or outside as a whole
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = out_row(a, e);
          if (row >= g.M) continue;
          vecT o;
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const float hv = (EPI == EPI_ELU_GRAD) ? g.H[(size_t)row * g.ldh + col0 + b] : 0.f;
            o[b] = apply(acc[a][b][e], bv[b], hv);
          }
          *reinterpret_cast<vecT*>(Cb + (size_t)row * g.ldc + col0) = o;
        }
      }
    }
  }
  if (EPI == EPI_SLAB && !A_KM) {
    if (db_owner) {
#pragma unroll
      for (int c = 0; c < DBN; ++c) {
        const int col = tid + c * NT;
        if (col < BM && m0 + col < g.M) g.dbias[((size_t)split * g.db_parts + tile_n) * g.M + m0 + col] = dbacc[c];
      }
    }
  }
}

template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL>
__device__ __forceinline__ void hx_gemm_tile_sp(const GemmArgs& g, const int logical, float* __restrict__ lds) {
  const int tiles_mn = g.tiles_m * g.tiles_n;
  const int split = logical / tiles_mn;
  const int t = logical % tiles_mn;
  int k_begin = 0, k_end = g.K;
  if (EPI == EPI_SLAB) { k_begin = split * g.kchunk; k_end = min(g.K, k_begin + g.kchunk); }
  hx_gemm_tile_sp_ex<BM, BN, BKT, A_KM, B_KM, EPI, WM, WN, KFULL>(g, t / g.tiles_n, t % g.tiles_n, split, k_begin, k_end, lds);
}

template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL>
__global__ void __launch_bounds__(64 * WM * WN) hx_gemm_sp_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[GemmSp<BM, BN, BKT, A_KM, B_KM, WM, WN>::FLOATS];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  hx_gemm_tile_sp<BM, BN, BKT, A_KM, B_KM, EPI, WM, WN, KFULL>(g, logical, lds);
}

// The product on a SMALL fixed grid whose workgroups walk the tiles -- the rollout's background critic (hx_gemm_persistent_kernel of
// hx_gemm.h) on the slot-placed loop.  With WM x WN = 1 x 2 (two waves) and one workgroup per CU its waves take two SIMDs of every CU and
// leave the other two free: the env step's 512-register waves then sit two per CU as when they have the chip to themselves, instead
// of four per CU on the half of the chip that four-wave critic workgroups leave.
template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL>
__global__ void __launch_bounds__(64 * WM * WN) hx_gemm_sp_persistent_kernel(GemmArgs g, int total_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[GemmSp<BM, BN, BKT, A_KM, B_KM, WM, WN>::FLOATS];
  int paused = 0;
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    hx_gemm_tile_sp_ex<BM, BN, BKT, A_KM, B_KM, EPI, WM, WN, KFULL, true>(g, t / g.tiles_n, t % g.tiles_n, 0, 0, g.K, lds, SP_WHOLE, nullptr, &paused);
    __syncthreads();
  }
}

// several products in one launch: see hx_gemm_group_kernel (same block -> work map)
template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL>
__global__ void __launch_bounds__(64 * WM * WN) hx_gemm_sp_group_kernel(GemmGroup G) {
  __shared__ __attribute__((aligned(16))) float lds[GemmSp<BM, BN, BKT, A_KM, B_KM, WM, WN>::FLOATS];
  int pi, logical;
  hx_group_pick(G, pi, logical);
  if (pi < 0) return;
  GemmArgs g = G.p[0];
#pragma unroll
  for (int i = 1; i < HX_GROUP_MAX; ++i) if (pi == i) g = G.p[i];
  hx_gemm_tile_sp<BM, BN, BKT, A_KM, B_KM, EPI, WM, WN, KFULL>(g, logical, lds);
}

// ---- forward / input-gradient products at ONE workgroup per CU: a stream-K cut ---------------------------------
// 256 x 256 tiles at one wave per SIMD run the K loop at 0.95 of the pipe (above), but 61 440 rows x {768, 512} columns are 1200 such
// tiles of two lengths on 256 CUs: 4.7 rounds, and a launch ends with the last round's slowest tile (measured 1136 us against 1075 us
// for the 128 x 128 kernel at three workgroups per CU).  So the launch is cut by WORK, not by tiles: the K tiles of all output tiles of
// all members form one sequence (member-major, tile-major, k-minor) and workgroup b of G takes the b-th G-th of it.  A range covers a
// few whole tiles plus, at most, the TAIL of a tile at its start and the HEAD of one at its end (a range is never shorter than a
// tile's K loop: the host checks).  The workgroup that computes a tile's head owns the tile: it does that segment LAST in its range,
// adds the partial sums the next workgroup left for it -- computed FIRST in that workgroup's range, so they are long there -- and runs
// the epilogue.  Hand-over: cdna_hip_programming.md Guideline 16 (every storing wave drains, barrier, one lane's agent-scope release,
// a relaxed flag store; the owner's lane polls relaxed with s_sleep and a bound, one agent-scope acquire, barrier, plain loads).  A
// flag holds the launch's epoch (a counter the host increments per launch), so nothing is re-zeroed between launches.  Sums: a split
// tile adds (head partial) + (tail partial) instead of one chain over k -- fixed by the shapes alone, so results are reproducible run to
// run, and equal to hx_gemm.h's to fp32 round-off on the ~G split tiles and bit for bit on all others.
struct GemmSk {
  GemmGroup G;                        // members (tiles_m / tiles_n filled; first[] unused)
  int nk[HX_GROUP_MAX];               // K tiles per output tile of a member
  long long it0[HX_GROUP_MAX + 1];    // first K tile of a member in the launch's sequence
  float* partial;                     // [workgroups][accumulator registers][threads]
  unsigned* flags;                    // [workgroups]: epoch of the last launch in which workgroup b published its partial tile
  unsigned epoch;
  int* err;                           // set to 1 if a wait ran out of its budget (the result is then wrong; the host checks)
};
template <int BM, int BN, int BKT, bool A_KM, bool B_KM, int EPI, int WM, int WN, bool KFULL>
__global__ void __launch_bounds__(64 * WM * WN) hx_gemm_sk_kernel(GemmSk S) {
  using P = GemmSp<BM, BN, BKT, A_KM, B_KM, WM, WN>;
  __shared__ __attribute__((aligned(16))) float lds[P::FLOATS];
  constexpr size_t SLOT = (size_t)P::TM * P::TN * 16 * P::NT;          // floats of one workgroup's partial tile
  const int b = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
  const long long total = S.it0[S.G.n];
  const long long begin = total * b / nwg, end = total * (b + 1) / nwg;
  long long it = begin;
  while (it < end) {
    int m = 0;
#pragma unroll
    for (int i = 1; i < HX_GROUP_MAX; ++i) if (i < S.G.n && it >= S.it0[i]) m = i;
    GemmArgs g = S.G.p[0]; int nk = S.nk[0]; long long base = S.it0[0];
#pragma unroll
    for (int i = 1; i < HX_GROUP_MAX; ++i) if (m == i) { g = S.G.p[i]; nk = S.nk[i]; base = S.it0[i]; }
    const long long rel = it - base;
    const int tile = (int)(rel / nk), ks = (int)(rel % nk);
    const int ke = (int)((long long)nk < ks + (end - it) ? (long long)nk : ks + (end - it));
    const int tile_m = tile / g.tiles_n, tile_n = tile % g.tiles_n;
    const int k_begin = ks * BKT, k_end = min(g.K, ke * BKT);
    int mode = SP_WHOLE;
    float* part = nullptr;
    if (ks > 0) { mode = SP_PART_OUT; part = S.partial + (size_t)b * SLOT; }
    else if (ke < nk) {
      // owner of a tile whose tail the next workgroup computed at the start of its range
      mode = SP_PART_IN; part = S.partial + (size_t)(b + 1) * SLOT;
      if (tid == 0) {
        int budget = 1 << 22;                    // x s_sleep(32): seconds; never reached unless the next workgroup never ran
        while (__hip_atomic_load(S.flags + b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S.epoch && --budget > 0) __builtin_amdgcn_s_sleep(32);
        if (budget <= 0) atomicExch(S.err, 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    hx_gemm_tile_sp_ex<BM, BN, BKT, A_KM, B_KM, EPI, WM, WN, KFULL>(g, tile_m, tile_n, 0, k_begin, k_end, lds, mode, part);
    if (mode == SP_PART_OUT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(S.flags + b, S.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();          // the next segment's first LDS stores must not overtake this one's last fragment reads
    it += ke - ks;
  }
}

// ---- the weight gradients of a minibatch at ONE workgroup per CU ---------------------------------------------
// Measured (profiles/r04_b_gemm_lab.txt): with the slot-placed loop a single wave per SIMD holding a 128 x 128 output tile (256
// accumulator registers) keeps the matrix pipe busier than three co-resident waves with 64 x 64 tiles each (0.92-0.96 of the
// pipe's cycles against 0.86-0.90): per MFMA it issues half the fragment reads and a quarter of the staging instructions, and
// nobody else's non-matrix instructions sit between its MFMAs.  One workgroup per CU means the launch must be cut into exactly
// as many equal pieces as there are CUs; the six dZ^T X products of the hector networks come in five shapes, so a member of the
// launch carries a tile SHAPE (one of a fixed menu, all instantiated in this kernel), and its own split count, chosen by the host
// planner (hx_wgrad_plan.h) so that  (MFMA tiles per wave) x (rows per split)  is the same for every workgroup.
//   shape  workgroup tile   waves   per wave            used for
//   0      256 x 256        2 x 2   128 x 128 (16)      out, in multiples of 256 (768 x 1024 of the critic's input layer, 256 x 768, 256 x 512)
//   1      512 x 128        4 x 1   128 x 128 (16)      512 x 616 (in = 4.8 x 128: five tiles, 3.75 % padding)
//   2      128 x 256        1 x 4   128 x  64  (8)      the two 128 x 256 layers
//   3      128 x 128        2 x 2    64 x  64  (4)      anything else (edges guarded as in hx_gemm.h)
//   4      512 x  32        4 x 1   128 x  32  (4)      column strips: the last 28 of the critic's 1052 input columns, so that the
//                                                        1024 before them are whole 256-wide tiles (a fifth 256-wide tile: 17.8 % padding)
#define HX_WSHAPES 5
struct WgradShape { int bm, bn, units; };        // units = 32x32 MFMA tiles per wave: a workgroup's time is units x rows-per-split
static const WgradShape HX_WSHAPE[HX_WSHAPES] = {{256, 256, 16}, {512, 128, 16}, {128, 256, 8}, {128, 128, 4}, {512, 32, 4}};
struct WgradMulti { GemmGroup G; int shape[HX_GROUP_MAX]; };
static constexpr int hx_wgrad_multi_lds_floats() {
  int m = GemmSp<256, 256, 16, false, false, 2, 2>::FLOATS;
  if (GemmSp<512, 128, 16, false, false, 4, 1>::FLOATS > m) m = GemmSp<512, 128, 16, false, false, 4, 1>::FLOATS;
  if (GemmSp<128, 256, 16, false, false, 1, 4>::FLOATS > m) m = GemmSp<128, 256, 16, false, false, 1, 4>::FLOATS;
  if (GemmSp<128, 128, 16, false, false, 2, 2>::FLOATS > m) m = GemmSp<128, 128, 16, false, false, 2, 2>::FLOATS;
  if (GemmSp<512, 32, 16, false, false, 4, 1>::FLOATS > m) m = GemmSp<512, 32, 16, false, false, 4, 1>::FLOATS;
  return m;
}
__global__ void __launch_bounds__(256) hx_wgrad_multi_kernel(WgradMulti W) {
  __shared__ __attribute__((aligned(16))) float lds[hx_wgrad_multi_lds_floats()];
  const GemmGroup& G = W.G;
  int pi, logical;
  hx_group_pick(G, pi, logical);
  if (pi < 0) return;
  GemmArgs g = G.p[0]; int shape = W.shape[0];
#pragma unroll
  for (int i = 1; i < HX_GROUP_MAX; ++i) if (pi == i) { g = G.p[i]; shape = W.shape[i]; }
  if (shape == 0) hx_gemm_tile_sp<256, 256, 16, false, false, EPI_SLAB, 2, 2, true>(g, logical, lds);
  else if (shape == 1) hx_gemm_tile_sp<512, 128, 16, false, false, EPI_SLAB, 4, 1, true>(g, logical, lds);
  else if (shape == 2) hx_gemm_tile_sp<128, 256, 16, false, false, EPI_SLAB, 1, 4, true>(g, logical, lds);
  else if (shape == 3) hx_gemm_tile_sp<128, 128, 16, false, false, EPI_SLAB, 2, 2, true>(g, logical, lds);
  else hx_gemm_tile_sp<512, 32, 16, false, false, EPI_SLAB, 4, 1, true>(g, logical, lds);
}
