// hx_gemm_bf16.h -- mixed-precision GEMM for the learner's forward and dgrad products (BASELINE config 4:
// "bf16 MLP on MFMA"): operands stay fp32 in HBM, are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on
// their way into LDS, multiplied on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate) and accumulated in fp32; the
// epilogue and the stored result are fp32.  Master weights, Adam and the weight-gradient product remain fp32
// (hx_gemm.h), so this changes the precision of activations / input gradients only.
//
//   C[M,N] = epi( A[M,K] . B[N,K]^T )       A, B K-major (row = output index, K contiguous)
//     FWD    A = X,  B = W              epi = bias + ELU
//     DGRAD  A = dZ, B = W^T (the learner keeps an fp32 transposed copy of the hidden weights)   epi = * elu'(H)
//
// Tile 128 x 128 x 64 per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA tiles).  The LDS image is
// [row][64 + 8] bf16 = 144-byte rows -- byte for byte the geometry of the fp32 kernel's [row][32 + 4] float tile, so
// the same conflict-free ds_write_b64 / ds_read_b128 pattern applies: lane (r = lane & 31, h = lane >> 5) reads the
// 16 bytes  k = 16 s + 8 h .. + 7  of its row for MFMA step s, which is exactly the 32x32x16 operand layout.
// With fp32 operands in HBM these products are memory-bound (a 61 440 x 616 -> 512 layer moves 277 MB for 38.8 GFLOP);
// the roofline that applies is HBM / L2 bandwidth, not the 2.5 PFLOP/s bf16 peak.
#pragma once
#include <hip/hip_runtime.h>
#include "hx_gemm.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define HXB_BK 64
#define HXB_LD (HXB_BK + 8)        // bf16 elements per LDS row

template <int EPI>
__global__ void __launch_bounds__(256) hx_gemm_bf16_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128;
  constexpr int TILE = (BM + BN) * HXB_LD;             // bf16 elements per buffer
  constexpr int LOADS = BM * HXB_BK / 4 / 256;         // float4 per thread per operand per tile (= 8)
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * TILE];

  const int nwg = gridDim.x, bid = blockIdx.x;
  int logical;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tile_m = logical / g.tiles_n, tile_n = logical % g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = (g.K + HXB_BK - 1) / HXB_BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r32 = lane & 31;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[LOADS], rb[LOADS];
  // Interior tiles (all of them at the production sizes) load through one base pointer per operand plus uniform offsets:
  // no per-tile 64-bit address arithmetic and no divergent regions in the MFMA loop (hx_gemm.h, profiles/r01_n_mfma_probe.txt)
  const bool fast_mn = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  const int row_t = tid / (HXB_BK / 4), k4_t = tid % (HXB_BK / 4);
  const float* pa0 = g.A + (size_t)(fast_mn ? m0 + row_t : 0) * g.lda + k4_t * 4;
  const float* pb0 = g.B + (size_t)(fast_mn ? n0 + row_t : 0) * g.ldb + k4_t * 4;
  constexpr int ROWS_PER_PASS = 256 / (HXB_BK / 4);
  const size_t sa = (size_t)ROWS_PER_PASS * g.lda, sb = (size_t)ROWS_PER_PASS * g.ldb;
  auto load_tile = [&](int kt) {
    const int k0 = kt * HXB_BK;
    if (fast_mn && k0 + HXB_BK <= g.K) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(pa0 + k0 + i * sa);
        rb[i] = *reinterpret_cast<const f32x4*>(pb0 + k0 + i * sb);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (HXB_BK / 4), k4 = idx % (HXB_BK / 4);
      const int gk = k0 + k4 * 4;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (m0 + row < g.M && gk < g.K) va = *reinterpret_cast<const f32x4*>(g.A + (size_t)(m0 + row) * g.lda + gk);
      if (n0 + row < g.N && gk < g.K) vb = *reinterpret_cast<const f32x4*>(g.B + (size_t)(n0 + row) * g.ldb + gk);
      ra[i] = va; rb[i] = vb;
    }
  };
  auto store_tile = [&](int buf) {
    __bf16* As = lds + buf * TILE;
    __bf16* Bs = As + BM * HXB_LD;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / (HXB_BK / 4), k4 = idx % (HXB_BK / 4);
      const bf16x4 pa = {(__bf16)ra[i][0], (__bf16)ra[i][1], (__bf16)ra[i][2], (__bf16)ra[i][3]};
      const bf16x4 pb = {(__bf16)rb[i][0], (__bf16)rb[i][1], (__bf16)rb[i][2], (__bf16)rb[i][3]};
      *reinterpret_cast<bf16x4*>(As + row * HXB_LD + k4 * 4) = pa;
      *reinterpret_cast<bf16x4*>(Bs + row * HXB_LD + k4 * 4) = pb;
    }
  };
  auto compute = [&](int buf) {
    const __bf16* As = lds + buf * TILE;
    const __bf16* Bs = As + BM * HXB_LD;
#pragma unroll
    for (int s = 0; s < HXB_BK / 16; ++s) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + i * 32 + r32) * HXB_LD + s * 16 + 8 * h);
        fb[i] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + i * 32 + r32) * HXB_LD + s * 16 + 8 * h);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  };

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1 < nk);
      if (more) load_tile(kt + 1);
      compute(kt & 1);
      if (more) store_tile((kt + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of v_mfma_f32_32x32x16_bf16 = that of the fp32 32x32x2 instruction
  //      (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); branch-free for interior tiles (hx_gemm.h)
  const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + wn * 64 + b * 32 + r32;
      const int row0 = m0 + wm * 64 + a * 32 + 4 * h;
      if (interior) {
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
        for (int e = 0; e < 16; ++e) g.C[(size_t)(row0 + (e & 3) + 8 * (e >> 2)) * g.ldc + col] = out[e];
      } else {
        if (col >= g.N) continue;
        float bv = 0.f;
        if (EPI == EPI_BIAS_ELU || EPI == EPI_BIAS) bv = g.bias[col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2);
          if (row >= g.M) continue;
          float v = acc[a][b][e];
          if (EPI == EPI_BIAS_ELU) v = hx_elu(v + bv);
          if (EPI == EPI_BIAS) v = v + bv;
          if (EPI == EPI_ELU_GRAD) {
            const float hh = g.H[(size_t)row * g.ldh + col];
            v = v * (hh > 0.f ? 1.f : hh + 1.f);
          }
          g.C[(size_t)row * g.ldc + col] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient product in the same mixed precision:  dW[M,N] = A[K,M]^T B[K,N]  (split-K, partial slabs)
//   A = dZ [rows][out]  (M-major),  B = X [rows][in]  (N-major), both fp32 in HBM, K = sample rows.
// The 32x32x16 operand wants 8 k-values of ONE row per lane, but here k is the slow index of both operands.  A dot
// product does not care in which order its k terms are visited, so the LDS image packs k PAIRS: word [p][m] holds
// (bf16 x[2p][m], bf16 x[2p+1][m]).  A thread that loaded the 4(k) x 4(m) block {4kb .. 4kb+3} x {4mb .. 4mb+3}
// (four coalesced float4 loads) owns two complete rows of four words and stores them with two ds_write_b128 --
// consecutive lanes write consecutive 16 bytes -- and a lane's fragment for MFMA step s is the four words
// [8s + 4h + q][row], q = 0..3 (two ds_read2_b32, consecutive lanes read consecutive words): no bank conflicts on
// either side, and A and B use the same k <-> (lane half, element) assignment.
// The bias gradient (column sums of dZ) is accumulated from the fp32 staging registers, not from the rounded tile.
__global__ void __launch_bounds__(256) hx_wgrad_bf16_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128, BKP = HXB_BK / 2;            // BKP = k pairs per tile
  constexpr int OPW = BKP * BM;                                  // words per operand tile
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 2 * OPW];     // 2 buffers x (A, B) = 64 KB

  const int nwg = gridDim.x, bid = blockIdx.x;
  int logical;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tiles_mn = g.tiles_m * g.tiles_n;
  const int split = logical / tiles_mn;
  const int t = logical % tiles_mn;
  const int tile_m = t / g.tiles_n, tile_n = t % g.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_begin = split * g.kchunk, k_end = min(g.K, k_begin + g.kchunk);
  const int nk = (k_end - k_begin + HXB_BK - 1) / HXB_BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r32 = lane & 31;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra[2][4], rb[2][4];                       // [block][k offset 0..3]
  const bool want_db = (g.dbias != nullptr) && (tile_n == 0);
  f32x4 dbacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  // whole tiles inside the matrix and the reduction range (every tile but the last of a split at the production sizes):
  // one base pointer per operand, uniform offsets, no divergent regions in the MFMA loop (hx_gemm.h)
  const bool fast_mn = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  const float* pa0 = g.A + (size_t)(k_begin + 4 * (tid >> 5)) * g.lda + (fast_mn ? m0 + 4 * (tid & 31) : 0);
  const float* pb0 = g.B + (size_t)(k_begin + 4 * (tid >> 5)) * g.ldb + (fast_mn ? n0 + 4 * (tid & 31) : 0);
  auto load_tile = [&](int kt) {
    const int k0 = k_begin + kt * HXB_BK;
    if (fast_mn && k0 + HXB_BK <= k_end) {
      const float* pa = pa0 + (size_t)(kt * HXB_BK) * g.lda;
      const float* pb = pb0 + (size_t)(kt * HXB_BK) * g.ldb;
#pragma unroll
      for (int i = 0; i < 2; ++i)               // idx = tid + 256 i  ->  kb = (tid >> 5) + 8 i: rows 32 i further down
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ra[i][j] = *reinterpret_cast<const f32x4*>(pa + (size_t)(32 * i + j) * g.lda);
          rb[i][j] = *reinterpret_cast<const f32x4*>(pb + (size_t)(32 * i + j) * g.ldb);
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int kb = idx >> 5, mb = idx & 31;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gk = k0 + 4 * kb + j;
        f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
        if (gk < k_end && m0 + 4 * mb < g.M) va = *reinterpret_cast<const f32x4*>(g.A + (size_t)gk * g.lda + m0 + 4 * mb);
        if (gk < k_end && n0 + 4 * mb < g.N) vb = *reinterpret_cast<const f32x4*>(g.B + (size_t)gk * g.ldb + n0 + 4 * mb);
        ra[i][j] = va; rb[i][j] = vb;
      }
    }
  };
  auto pack = [](float lo, float hi) -> unsigned {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<const unsigned*>(&p);
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto store_tile = [&](int buf) {
    unsigned* As = lds + buf * 2 * OPW;
    unsigned* Bs = As + OPW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int kb = idx >> 5, mb = idx & 31;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const f32x4 a0 = ra[i][2 * pr], a1 = ra[i][2 * pr + 1], b0 = rb[i][2 * pr], b1 = rb[i][2 * pr + 1];
        const u32x4 wa = {pack(a0[0], a1[0]), pack(a0[1], a1[1]), pack(a0[2], a1[2]), pack(a0[3], a1[3])};
        const u32x4 wb = {pack(b0[0], b1[0]), pack(b0[1], b1[1]), pack(b0[2], b1[2]), pack(b0[3], b1[3])};
        *reinterpret_cast<u32x4*>(As + (2 * kb + pr) * BM + 4 * mb) = wa;
        *reinterpret_cast<u32x4*>(Bs + (2 * kb + pr) * BN + 4 * mb) = wb;
      }
      if (want_db) dbacc[i] = dbacc[i] + ((ra[i][0] + ra[i][1]) + (ra[i][2] + ra[i][3]));
    }
  };
  auto compute = [&](int buf) {
    const unsigned* As = lds + buf * 2 * OPW;
    const unsigned* Bs = As + OPW;
#pragma unroll
    for (int s = 0; s < HXB_BK / 16; ++s) {
      union Frag { unsigned w[4]; bf16x8 v; } fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fa[i].w[q] = As[(8 * s + 4 * h + q) * BM + wm * 64 + i * 32 + r32];
          fb[i].w[q] = Bs[(8 * s + 4 * h + q) * BN + wn * 64 + i * 32 + r32];
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a].v, fb[b].v, acc[a][b], 0, 0, 0);
    }
  };

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1 < nk);
      if (more) load_tile(kt + 1);
      compute(kt & 1);
      if (more) store_tile((kt + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: partial slab of this split (fp32), same C/D layout as every other kernel here
  float* Cb = g.C + (size_t)split * g.M * g.ldc;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + wn * 64 + b * 32 + r32;
      if (col >= g.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < g.M) Cb[(size_t)row * g.ldc + col] = acc[a][b][e];
      }
    }
  if (want_db) {
    // 16 threads (the kb groups) hold partial sums for the same four columns: reduce through LDS, fixed order
    float* red = reinterpret_cast<float*>(lds);          // [16][128]; all tile reads are behind the last barrier
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int kb = idx >> 5, mb = idx & 31;
      *reinterpret_cast<f32x4*>(red + kb * BM + 4 * mb) = dbacc[i];
    }
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) sum += red[kb * BM + tid];
      g.dbias[(size_t)split * g.M + m0 + tid] = sum;
    }
  }
}
