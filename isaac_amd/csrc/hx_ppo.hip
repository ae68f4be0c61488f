// hx_ppo.hip -- PPO learner kernels + C ABI (include/hx_ppo.h).
//
// Restates the reference's humanoid/algo/ppo/{actor_critic,rollout_storage,ppo}.py (see hx_ppo.h for the
// function-by-function map).  The autograd graph is written out by hand; oracle/ppo.py is the CPU
// statement of the same formulas, pinned to the reference by tests/golden/ppo_*.npz.
//
// Data layout in HBM (fp32 unless noted):
//   params / grads / Adam m,v : one flat buffer each, per layer W[out][in_ld] (in_ld = in rounded up to 4,
//                               zero padded so every row is 16-byte aligned), then b[out]; std last.
//   rollout storage           : [T][N][ld] for obs / privileged obs, [T][N][A] actions / mu, [T][N] scalars.
//   minibatch workspace       : gathered rows + one activation buffer per hidden layer, [M][width].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/hx_ppo.h"
#include "../../include/hx_sim.h"
#include "../../include/hx_lab.h"
#include "hx_common.h"
#include <map>
#include "hx_gemm.h"
#include "hx_gemm_sp.h"
#include "hx_wgrad_plan.h"
#include "hx_gemm_bf16.h"

#define MAX_A 32
#ifndef HX_BK_UPD
#define HX_BK_UPD 16      // K-depth of the LDS tile for the 61 440-row update GEMMs
#endif
#ifndef HX_WGRAD_MIN_CHUNK
#define HX_WGRAD_MIN_CHUNK 256   // shortest K chunk of a split-K workgroup (rows)
#endif
#ifndef HX_BK_ROLL
#define HX_BK_ROLL 16     // K-depth for the 4096-row rollout GEMMs
#endif
static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ================================================================= small kernels
__device__ __forceinline__ void philox4p(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define LOG_SQRT_2PI 0.9189385332046727f

// Rollout head (ppo.py:91-101): mu = W4 h3a + b4, a = mu + std*eps, logp, V = w4c.h3c + b4c.
// 16 lanes per env row (each lane owns hw/16 consecutive k), partial dot products reduced with 4 xor-shuffles,
// so a 4096-env rollout step fills 256 workgroups instead of 16.
__global__ void __launch_bounds__(256) hx_act_head_kernel(const float* __restrict__ h3a, const float* __restrict__ h3c, int hw, int hwc,
                                                          const float* __restrict__ W4, const float* __restrict__ b4,
                                                          const float* __restrict__ W4c, const float* __restrict__ b4c,
                                                          const float* __restrict__ stdp, const float* __restrict__ eps,
                                                          int n, int A, uint32_t k0, uint32_t k1, uint32_t step, uint32_t row_base,
                                                          float* actions, float* mu_out, float* values, float* logp) {
  extern __shared__ float sm[];
  float* sW = sm;                 // [A][hw]
  float* sWc = sm + A * hw;       // [hwc]
  for (int i = threadIdx.x; i < A * hw; i += blockDim.x) sW[i] = W4[i];
  for (int i = threadIdx.x; i < hwc; i += blockDim.x) sWc[i] = W4c[i];
  __syncthreads();
  const int part = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int ec = e < n ? e : n - 1;            // keep every lane in the shuffles
  const int per = hw / 16, perc = hwc / 16;
  float mu[MAX_A];
  for (int j = 0; j < A; ++j) mu[j] = 0.f;
  float v = 0.f;
  const float* ha = h3a + (size_t)ec * hw + part * per;
  const float* hc = h3c + (size_t)ec * hwc + part * perc;
  for (int k = 0; k < per; k += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(ha + k);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = part * per + k + q;
      for (int j = 0; j < A; ++j) mu[j] = fmaf(x[q], sW[j * hw + kk], mu[j]);
    }
  }
  for (int k = 0; k < perc; k += 4) {
    const f32x4 y = *reinterpret_cast<const f32x4*>(hc + k);
#pragma unroll
    for (int q = 0; q < 4; ++q) v = fmaf(y[q], sWc[part * perc + k + q], v);
  }
  for (int o = 8; o > 0; o >>= 1) {
    for (int j = 0; j < A; ++j) mu[j] += __shfl_xor(mu[j], o);
    v += __shfl_xor(v, o);
  }
  if (part != 0 || e >= n) return;
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float m = mu[j] + b4[j];
    const float sg = m * 0.f + stdp[j];
    float z;
    if (eps) z = eps[(size_t)e * A + j];
    else {
      uint32_t o[4];
      philox4p(k0, k1, (uint32_t)e + row_base, step, (uint32_t)j, 7u, o);
      const float u1 = 1.0f - (float)(o[0] >> 8) * (1.0f / 16777216.0f);
      const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
      z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    const float a = m + sg * z;
    actions[(size_t)e * A + j] = a;
    mu_out[(size_t)e * A + j] = m;
    const float d = a - m;
    lp += -(d * d) / (2.0f * sg * sg) - logf(sg) - LOG_SQRT_2PI;
  }
  values[e] = v + b4c[0];
  logp[e] = lp;
}

// actor half of the rollout head: mu, sample, log-prob (16 lanes per row).  Values come from the deferred critic.
__global__ void __launch_bounds__(256) hx_actor_head_kernel(const float* __restrict__ h3a, int hw, const float* __restrict__ W4,
                                                            const float* __restrict__ b4, const float* __restrict__ stdp,
                                                            const float* __restrict__ eps, int n, int A, uint32_t k0, uint32_t k1,
                                                            uint32_t step, uint32_t row_base, float* actions, float* mu_out, float* logp) {
  extern __shared__ float sm[];
  float* sW = sm;
  for (int i = threadIdx.x; i < A * hw; i += blockDim.x) sW[i] = W4[i];
  __syncthreads();
  const int part = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int ec = e < n ? e : n - 1;
  const int per = hw / 16;
  float mu[MAX_A];
  for (int j = 0; j < A; ++j) mu[j] = 0.f;
  const float* ha = h3a + (size_t)ec * hw + part * per;
  for (int k = 0; k < per; k += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(ha + k);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = part * per + k + q;
      for (int j = 0; j < A; ++j) mu[j] = fmaf(x[q], sW[j * hw + kk], mu[j]);
    }
  }
  for (int o = 8; o > 0; o >>= 1)
    for (int j = 0; j < A; ++j) mu[j] += __shfl_xor(mu[j], o);
  if (part != 0 || e >= n) return;
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float m = mu[j] + b4[j];
    const float sg = m * 0.f + stdp[j];
    float z;
    if (eps) z = eps[(size_t)e * A + j];
    else {
      uint32_t o[4];
      philox4p(k0, k1, (uint32_t)e + row_base, step, (uint32_t)j, 7u, o);
      const float u1 = 1.0f - (float)(o[0] >> 8) * (1.0f / 16777216.0f);
      const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
      z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    const float a = m + sg * z;
    actions[(size_t)e * A + j] = a;
    mu_out[(size_t)e * A + j] = m;
    const float d = a - m;
    lp += -(d * d) / (2.0f * sg * sg) - logf(sg) - LOG_SQRT_2PI;
  }
  logp[e] = lp;
}


// ---------------------------------------------------------------------------------------------------------------
// Fused actor forward for the rollout (actor_critic.py:111-117 at M = num_envs): obs -> 3 x (Linear + ELU) -> Linear ->
// sample -> log-prob in ONE launch.  At 4096 rows the layer-by-layer GEMMs are latency-bound (a 64x128 tile walks its
// whole K loop alone on a CU), so here a workgroup owns only 16 rows -- 256 workgroups, one per CU -- keeps every
// activation in LDS (16 x (616+512+256+128) floats) and streams the weights from L2 straight into MFMA B fragments.
// v_mfma_f32_16x16x4_f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]; with the k-permutation used by
// hx_gemm.h a float4 per lane (k = 4*(l>>4) .. +3 of a 16-deep block) feeds 4 MFMAs.  C/D: col = l&15, row = 4*(l>>4)+reg.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define FA_ROWS 16      // rows of a fused-actor workgroup with one row tile (hx_actor_fused_kernel RT = 1)
// rows that live as windows of per-robot frame rings (single-frame observation storage): row r = base + off[r], elements
// [0, kz[r]) and [klim, ..) read as zero.  base == nullptr: ordinary rows.
struct FrameSrc { const float* base; const int* off; const int* kz; int klim; };
// operand rounding of the mixed-precision mode: fp32 -> bf16 (RNE) -> fp32, so that an fp32 MFMA on the rounded values
// reproduces what the bf16 matrix cores compute in the update (up to summation order)
__device__ __forceinline__ float hx_bf16r(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ f32x4v hx_bf16r4(f32x4v v) { return (f32x4v){hx_bf16r(v[0]), hx_bf16r(v[1]), hx_bf16r(v[2]), hx_bf16r(v[3])}; }

// Weight stream of the fused actor in FRAGMENT ORDER (hx_actor_pack_kernel below): for every 16-column tile nt and 16-deep
// k block kb the 64 float4 that the 64 lanes feed to four MFMAs lie in one contiguous KB,
//   P[((nt * nkb + kb) * 64 + lane) * 4 + j] = W[nt * 16 + (lane & 15)][kb * 16 + 4 * (lane >> 4) + j]      (0 beyond K),
// so a wave's load instruction reads 1 KB of consecutive addresses instead of 16 rows x 64 B, and a wave walks its tile's
// nkb KB front to back.  The row-major form kept the texture addresser busy half the time on 16 lines per instruction
// (profiles/r01_g_actor_ring.txt).  Same operands in the same MFMAs: outputs are bitwise those of the row-major stream.
__global__ void __launch_bounds__(256) hx_actor_pack_kernel(const float* __restrict__ W, int N, int K, int ldw, float* __restrict__ P) {
  const int nkb = (K + 15) / 16;
  const size_t total = (size_t)(N / 16) * nkb * 256;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    const size_t blk = i >> 8;
    const int kb = (int)(blk % nkb), nt = (int)(blk / nkb);
    const int k = kb * 16 + 4 * (lane >> 4) + j;
    P[i] = (k < K) ? W[(size_t)(nt * 16 + (lane & 15)) * ldw + k] : 0.f;
  }
}

// First two k blocks of a wave's weight stream, requested ahead of the barrier that precedes the layer: the L2 round trip
// then runs while the workgroup is still staging rows / finishing the previous layer.
template <int NT>
__device__ __forceinline__ void fa_prefetch(const float* __restrict__ W, int K, int n_wave0, int lane, f32x4v* b0, f32x4v* b1) {
  const int nkb = (K + 15) / 16;
  const f32x4v* __restrict__ Wp = reinterpret_cast<const f32x4v*>(W) + (size_t)(n_wave0 >> 4) * nkb * 64 + lane;
#pragma unroll
  for (int t = 0; t < NT; ++t) { b0[t] = Wp[((size_t)t * nkb) * 64]; b1[t] = Wp[((size_t)t * nkb + min(1, nkb - 1)) * 64]; }
}

// RT = 16-row tiles of the workgroup (1 or 2): every weight fragment streamed from L2 feeds RT MFMAs
template <int NT, bool BF, bool ROUND_OUT, int RT = 1>   // NT = 16-column tiles per wave; BF: round the weight operand; ROUND_OUT: round what is stored
__device__ __forceinline__ void fa_layer(const float* __restrict__ Xs, int ldx, int K, const float* __restrict__ W, int ldw,
                                         const float* __restrict__ bias, float* __restrict__ Hs, int ldh, int n_wave0, int lane,
                                         f32x4v* b0, f32x4v* b1) {      // b0 / b1: blocks 0 and 1, already requested (fa_prefetch)
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4v acc[RT][NT];
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[q][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  const int nkb = (K + 15) / 16;
  // weights are streamed from L2 with TWO k-blocks in flight per wave (one was L2-latency bound: 80 us per call).
  // The loop is written out for three named buffers on purpose: a generic register-ring version of the same
  // schedule compiled to 86 us instead of 74 us (profiles/r01_g_actor_ring.txt).
  f32x4v b2[NT];
  // Branch-free on purpose: with a guard around each load hipcc loses track of the VM counter across the divergent
  // regions and waits `vmcnt(0)` before every group of MFMAs, i.e. also for the blocks just requested (the stream
  // then pays a full L2 round trip every third block: 74-86 us per call).  Out-of-range blocks re-read the last valid
  // 16 bytes of the row instead; their A operand is zero (step() guards it), so they contribute nothing.
  // W is the packed stream of this layer (see hx_actor_pack_kernel); ldw is unused.  Out-of-range blocks re-read the last one.
  const f32x4v* __restrict__ Wp = reinterpret_cast<const f32x4v*>(W) + (size_t)(n_wave0 >> 4) * nkb * 64 + lane;
  auto loadB = [&](int kb, f32x4v* dst) {
    const int kc = min(kb, nkb - 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = Wp[((size_t)t * nkb + kc) * 64];
  };
  auto roundB = [&](f32x4v* b) {
    if (BF) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = hx_bf16r4(b[t]);
    }
  };
  auto step = [&](int kb, f32x4v* bc) {
    roundB(bc);
    const int k = kb * 16 + 4 * kq;
    f32x4v a[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) { a[q] = (f32x4v){0.f, 0.f, 0.f, 0.f}; if (k < K) a[q] = *reinterpret_cast<const f32x4v*>(Xs + (q * 16 + r16) * ldx + k); }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < RT; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][i], bc[t][i], acc[q][t], 0, 0, 0);
  };
  int kb = 0;
  for (; kb + 2 < nkb; kb += 3) {
    loadB(kb + 2, b2); step(kb, b0);
    loadB(kb + 3, b0); step(kb + 1, b1);
    loadB(kb + 4, b1); step(kb + 2, b2);
  }
  if (kb < nkb) step(kb, b0);
  if (kb + 1 < nkb) step(kb + 1, b1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n_wave0 + t * 16 + r16;
    const float bv = bias[col];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float o = hx_elu(acc[q][t][r] + bv);
        Hs[(q * 16 + kq * 4 + r) * ldh + col] = ROUND_OUT ? hx_bf16r(o) : o;
      }
  }
}

// The same layer with D weight blocks in flight per wave (HX_ACTOR_DEPTH; fa_layer above is D = 2 written out by hand): a ring of
// D + 1 block buffers with STATIC indices only (the k loop advances D + 1 blocks per trip and its body is unrolled), blocks past the
// end re-read the last one and meet a zero A operand.
template <int NT, int D>
__device__ __forceinline__ void fa_prefetch_d(const float* __restrict__ W, int K, int n_wave0, int lane, f32x4v (*buf)[NT]) {
  const int nkb = (K + 15) / 16;
  const f32x4v* __restrict__ Wp = reinterpret_cast<const f32x4v*>(W) + (size_t)(n_wave0 >> 4) * nkb * 64 + lane;
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int t = 0; t < NT; ++t) buf[d][t] = Wp[((size_t)t * nkb + min(d, nkb - 1)) * 64];
}
template <int NT, bool BF, bool ROUND_OUT, int RT, int D>
__device__ __forceinline__ void fa_layer_d(const float* __restrict__ Xs, int ldx, int K, const float* __restrict__ W, const float* __restrict__ bias,
                                           float* __restrict__ Hs, int ldh, int n_wave0, int lane, f32x4v (*buf)[NT]) {
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4v acc[RT][NT];
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[q][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  const int nkb = (K + 15) / 16;
  const f32x4v* __restrict__ Wp = reinterpret_cast<const f32x4v*>(W) + (size_t)(n_wave0 >> 4) * nkb * 64 + lane;
  for (int kb = 0; kb < nkb; kb += D + 1) {
#pragma unroll
    for (int u = 0; u <= D; ++u) {
      {                                               // request block kb + u + D into the buffer that block kb + u - 1 has left
        const int kc = min(kb + u + D, nkb - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) buf[(u + D) % (D + 1)][t] = Wp[((size_t)t * nkb + kc) * 64];
      }
      f32x4v* bc = buf[u];
      if (BF) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bc[t] = hx_bf16r4(bc[t]);
      }
      const int k = (kb + u) * 16 + 4 * kq;
      f32x4v a[RT];
#pragma unroll
      for (int q = 0; q < RT; ++q) { a[q] = (f32x4v){0.f, 0.f, 0.f, 0.f}; if (k < K) a[q] = *reinterpret_cast<const f32x4v*>(Xs + (q * 16 + r16) * ldx + k); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < RT; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][i], bc[t][i], acc[q][t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n_wave0 + t * 16 + r16;
    const float bv = bias[col];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float o = hx_elu(acc[q][t][r] + bv);
        Hs[(q * 16 + kq * 4 + r) * ldh + col] = ROUND_OUT ? hx_bf16r(o) : o;
      }
  }
}

// RT = 16-row tiles per workgroup.  RT = 2 (32 rows, HX_ACTOR_ROWS=32, an experiment): half as many workgroups stream the 1.9 MB of
// weights out of the L2 and every fragment feeds two MFMAs; measured slower (fa_row_tiles).  The activations then share two LDS
// buffers: X and H2 in one, H1 and H3 in the other (a layer's input is dead once its output is complete).
template <bool BF, int NW, int RT = 1, int D = 0>   // NW waves per workgroup (4 or 8): each owns 1/NW of a layer's output columns; D > 0: fa_layer_d
__global__ void __launch_bounds__(64 * NW) hx_actor_fused_kernel(const float* __restrict__ obs, int obs_ld, int n,
                                                             const float* __restrict__ W1, const float* __restrict__ b1, int K1, int N1,
                                                             const float* __restrict__ W2, const float* __restrict__ b2, int N2,
                                                             const float* __restrict__ W3, const float* __restrict__ b3, int N3,
                                                             const float* __restrict__ W4, const float* __restrict__ b4,
                                                             const float* __restrict__ stdp, const float* __restrict__ eps, int A,
                                                             uint32_t k0, uint32_t k1, uint32_t step, uint32_t row_base,
                                                             float* actions, float* mu_out, float* logp, FrameSrc fsrc, hx_step_book book, int book_valid,
                                                             int* pause) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  // the background critic's persistent workgroups sleep while this count is up (hx_gemm.h GemmArgs::pause): the actor is on
  // the rollout's critical path and shares half the CUs with them
  if (pause != nullptr && threadIdx.x == 0) {
    atomicAdd(pause, 1);
    // the stacking launch before this one raised the count for itself (hx_sim_set_pause_word): taken back here, after our own +1
    if (blockIdx.x == 0 && atomicExch(pause + 1, 0) == 1) atomicSub(pause, 1);
  }
#ifdef HX_ACTOR_PROF
  // phase stamps of workgroup b (100 MHz ticks): pause[16 + 16 b + i]; tools/actor_prof.py
#define HX_AST(i) do { if (pause != nullptr && threadIdx.x == 0 && blockIdx.x < 4096) reinterpret_cast<long long*>(pause + 16)[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define HX_AST(i) do { } while (0)
#endif
  HX_AST(0);
  const int ldx = K1 + 4, ld1 = N1 + 4, ld2 = N2 + 4, ld3 = N3 + 4;
  constexpr int ROWS = 16 * RT;
  constexpr int W4_PER = (MAX_A * 128 + 64 * NW - 1) / (64 * NW);      // head weights per thread (A <= MAX_A actions x N3 = 128 inputs)
  float* Xs = fsm;
  float* W4s = fsm;                          // RT = 1: over X, which is dead once layer 1 is complete
  float* H1 = Xs + ROWS * (ldx > ld2 ? ldx : ld2);
  float* H2 = (RT == 1) ? H1 + ROWS * ld1 : Xs;              // RT = 2: H2 over X, H3 over H1
  float* H3 = (RT == 1) ? H2 + ROWS * ld2 : H1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * ROWS;
  f32x4v p1a[32 / NW], p1b[32 / NW], p2a[16 / NW], p2b[16 / NW], p3a[8 / NW], p3b[8 / NW];
  constexpr int DD = D > 0 ? D : 1;
  f32x4v q1[DD + 1][32 / NW], q2[DD + 1][16 / NW], q3[DD + 1][8 / NW];
  if (D > 0) fa_prefetch_d<32 / NW, DD>(W1, K1, wave * (512 / NW), lane, q1);
  else fa_prefetch<32 / NW>(W1, K1, wave * (512 / NW), lane, p1a, p1b);
  if (fsrc.base != nullptr) {
    // Single-frame storage (include/hx_sim.h): the 16 rows are windows of the robots' frame rings.  Row start and first valid
    // element come from the tables; elements before it (frames older than the robot's last reset) and the padding read as
    // zero.  Row starts are only float-aligned (41-wide frames), so the loads are scalar, coalesced along the row.
    // 64 * NW / 16 threads per row.  Row starts are only float-aligned (41-wide frames), so a thread loads ALIGNED float4 from the
    // row start rounded down -- covering elements -m .. K1 + 3 - m of the row, m = the start's misalignment in floats -- and
    // scatters the four values to their columns in LDS: a quarter of the load instructions of an element-wise copy, no split
    // 16-byte accesses, no index division.  All loads of a thread are independent of each other.
    constexpr int TPR = 64 * NW / ROWS;
    const int r = tid / TPR, q0 = tid % TPR, gr = min(row0 + r, n - 1);
    const int off = fsrc.off[gr], kz = fsrc.kz[gr], m = off & 3;
    const f32x4v* src = reinterpret_cast<const f32x4v*>(fsrc.base + (off - m));
    float* xr = Xs + r * ldx;
    for (int q = q0; 4 * q - m < K1; q += TPR) {
      const f32x4v v = src[q];                                       // always inside the ring (+ slack): no guard around the load
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = 4 * q + c - m;
        if (k >= 0 && k < K1) { const float o = (k >= kz && k < fsrc.klim) ? v[c] : 0.f; xr[k] = BF ? hx_bf16r(o) : o; }
      }
    }
    // bookkeeping of the env step that produced these rows (it needs that launch's total reset count, hx_common.h)
    if (book_valid) {
      if (tid < ROWS && row0 + tid < n) hx_step_book_row(book, row0 + tid);
      if (blockIdx.x == 0 && tid >= 64 && tid < 128) hx_step_book_global(book, tid - 64);
    }
  } else {
    // stage the 16 observation rows (coalesced float4; rows past n read row n-1 and are discarded at the end)
    for (int i = tid; i < ROWS * (K1 / 4); i += 64 * NW) {
      const int r = i / (K1 / 4), c4 = i % (K1 / 4);
      const int gr = min(row0 + r, n - 1);
      const f32x4v v = *reinterpret_cast<const f32x4v*>(obs + (size_t)gr * obs_ld + c4 * 4);
      *reinterpret_cast<f32x4v*>(Xs + r * ldx + c4 * 4) = BF ? hx_bf16r4(v) : v;
    }
  }
  __syncthreads();
  HX_AST(1);
  // this thread's head bias and sigma, fetched behind the layers (thread i of the head owns (row i / A, action i % A))
  const float hb4 = (tid < ROWS * A) ? b4[tid % A] : 0.f, hsg = (tid < ROWS * A) ? stdp[tid % A] : 0.f;
  if (D > 0) {
    fa_prefetch_d<16 / NW, DD>(W2, N1, wave * (256 / NW), lane, q2);
    fa_layer_d<32 / NW, BF, BF, RT, DD>(Xs, ldx, K1, W1, b1, H1, ld1, wave * (512 / NW), lane, q1);
    __syncthreads();
    fa_prefetch_d<8 / NW, DD>(W3, N2, wave * (128 / NW), lane, q3);
    fa_layer_d<16 / NW, BF, BF, RT, DD>(H1, ld1, N1, W2, b2, H2, ld2, wave * (256 / NW), lane, q2);
    __syncthreads();
    fa_layer_d<8 / NW, BF, false, RT, DD>(H2, ld2, N2, W3, b3, H3, ld3, wave * (128 / NW), lane, q3);
    __syncthreads();
  } else {
  fa_prefetch<16 / NW>(W2, N1, wave * (256 / NW), lane, p2a, p2b);
  fa_layer<32 / NW, BF, BF, RT>(Xs, ldx, K1, W1, K1, b1, H1, ld1, wave * (512 / NW), lane, p1a, p1b);      // 615(616) -> 512
  __syncthreads();
  HX_AST(2);
  fa_prefetch<8 / NW>(W3, N2, wave * (128 / NW), lane, p3a, p3b);
  // the head's weights (A x N3 floats) on their way to LDS behind layer 2: the head then reads both operands of its dot products from
  // LDS.  With W4 read from global memory inside the head's serial k loop the head took 5.9 us on average and up to 17 us in the
  // slowest workgroup -- and a launch ends with its slowest workgroup (profiles/r04_t_actor_rows.txt, tools/actor_prof.py)
  float w4r[W4_PER];
#pragma unroll
  for (int q = 0; q < W4_PER; ++q) { const int i = tid + q * 64 * NW; w4r[q] = (i < A * N3) ? W4[i] : 0.f; }

  fa_layer<16 / NW, BF, BF, RT>(H1, ld1, N1, W2, N1, b2, H2, ld2, wave * (256 / NW), lane, p2a, p2b);      // 512 -> 256
  if (RT == 1) {
#pragma unroll
    for (int q = 0; q < W4_PER; ++q) { const int i = tid + q * 64 * NW; if (i < A * N3) W4s[i] = w4r[q]; }
  }
  __syncthreads();
  HX_AST(3);
  fa_layer<8 / NW, BF, false, RT>(H2, ld2, N2, W3, N2, b3, H3, ld3, wave * (128 / NW), lane, p3a, p3b);    // 256 -> 128; the head reads H3 unrounded, like the update's fp32 loss head
  __syncthreads();
  HX_AST(4);
  }
  // head: mu[r][j] = W4[j] . H3[r] + b4[j]; one thread per (row, action) also samples its action and leaves its
  // log-prob term in LDS; the row's thread then adds the terms in action order (the order of the serial loop it replaces)
  float* sTerm = (RT == 1) ? H1 : Xs;       // dead by now: layer-1 activations (RT = 1); X / H2 (RT = 2, where H1's place holds H3)
  for (int i = tid; i < ROWS * A; i += 64 * NW) {
    const int r = i / A, j = i % A;
    float m = 0.f;
    if (RT == 1 && D == 0) { for (int k = 0; k < N3; ++k) m = fmaf(H3[r * ld3 + k], W4s[j * N3 + k], m); }
    else { for (int k = 0; k < N3; ++k) m = fmaf(H3[r * ld3 + k], W4[j * N3 + k], m); }
    const bool pre = (RT == 1 && D == 0 && ROWS * A <= 64 * NW);      // one pass: i == tid, bias and sigma were fetched behind layer 1
    m += pre ? hb4 : b4[j];
    const int e = row0 + r;
    if (e < n) {
      const float sg = m * 0.f + (pre ? hsg : stdp[j]);
      float z;
      if (eps) z = eps[(size_t)e * A + j];
      else {
        uint32_t o[4];
        philox4p(k0, k1, (uint32_t)e + row_base, step, (uint32_t)j, 7u, o);
        const float u1 = 1.0f - (float)(o[0] >> 8) * (1.0f / 16777216.0f);
        const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
        z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
      }
      const float a = m + sg * z;
      actions[(size_t)e * A + j] = a;
      mu_out[(size_t)e * A + j] = m;
      const float d = a - m;
      sTerm[r * MAX_A + j] = -(d * d) / (2.0f * sg * sg) - logf(sg) - LOG_SQRT_2PI;
    }
  }
  __syncthreads();
  HX_AST(5);
  if (tid < ROWS && row0 + tid < n) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) lp += sTerm[tid * MAX_A + j];
    logp[row0 + tid] = lp;
  }
  HX_AST(6);
  if (pause != nullptr && tid == 0) atomicSub(pause, 1);
}

// critic head only (bootstrap value of compute_returns, ppo.py:116)
__global__ void __launch_bounds__(256) hx_value_head_kernel(const float* __restrict__ h3c, int hw, const float* __restrict__ W4c,
                                                            const float* __restrict__ b4c, int n, float* values) {
  const int part = threadIdx.x & 15;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int ec = e < n ? e : n - 1;
  const int per = hw / 16;
  float v = 0.f;
  const float* hc = h3c + (size_t)ec * hw + part * per;
  for (int k = 0; k < per; ++k) v = fmaf(hc[k], W4c[part * per + k], v);
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (part == 0 && e < n) values[e] = v + b4c[0];
}

// actor head only (act_inference, actor_critic.py:122-124)
__global__ void __launch_bounds__(256) hx_mean_head_kernel(const float* __restrict__ h3a, int hw, const float* __restrict__ W4,
                                                           const float* __restrict__ b4, int n, int A, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * A) return;
  const int e = i / A, j = i % A;
  float m = 0.f;
  for (int k = 0; k < hw; ++k) m = fmaf(h3a[(size_t)e * hw + k], W4[j * hw + k], m);
  out[i] = m + b4[j];
}

// process_env_step (ppo.py:103-113): store reward / done / time-out flag.  The bootstrap  r += gamma * V * time_out
// (ppo.py:107-108) is applied by the GAE kernel, because the critic that produces V runs deferred, in large batches
// beside the env-step kernels (see act_impl).
__global__ void hx_process_step_kernel(const float* __restrict__ rew, const unsigned char* __restrict__ dones,
                                       const unsigned char* __restrict__ timeouts, int n, float* rew_out,
                                       unsigned char* done_out, unsigned char* timeout_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  rew_out[e] = rew[e];
  done_out[e] = dones[e] ? 1 : 0;
  timeout_out[e] = (timeouts && timeouts[e]) ? 1 : 0;
}

// GAE (rollout_storage.py:122-132) + first pass of the advantage moments
__global__ void __launch_bounds__(256) hx_gae_kernel(float* __restrict__ rewards, const unsigned char* __restrict__ dones,
                                                     const unsigned char* __restrict__ timeouts,
                                                     const float* __restrict__ values, const float* __restrict__ last_values,
                                                     int T, int n, float gamma, float lam, float* returns, float* adv_raw,
                                                     double* moments) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0, s2 = 0.0;
  if (e < n) {
    float adv = 0.f;
    for (int t = T - 1; t >= 0; --t) {
      const float nv = (t == T - 1) ? last_values[e] : values[(size_t)(t + 1) * n + e];
      const float nt = 1.0f - (float)dones[(size_t)t * n + e];
      const float v = values[(size_t)t * n + e];
      const float r = rewards[(size_t)t * n + e] + gamma * (v * (float)timeouts[(size_t)t * n + e]);   // ppo.py:107-108
      rewards[(size_t)t * n + e] = r;
      const float delta = r + nt * gamma * nv - v;
      adv = delta + nt * gamma * lam * adv;
      const float ret = adv + v;
      returns[(size_t)t * n + e] = ret;
      const float a = ret - v;
      adv_raw[(size_t)t * n + e] = a;
      s += (double)a; s2 += (double)a * (double)a;
    }
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&moments[0], sh[0][0]);
    atomicAdd(&moments[1], sh[1][0]);
    if (blockIdx.x == 0) atomicAdd(&moments[2], (double)T * (double)n);
  }
}

// (A - mean) / (std_unbiased + 1e-8)   (rollout_storage.py:135-136)
__global__ void hx_adv_normalize_kernel(const float* __restrict__ adv_raw, const double* __restrict__ moments, size_t count, float* adv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double nn = moments[2];
  const double mean = moments[0] / nn;
  double var = (moments[1] - nn * mean * mean) / (nn - 1.0);
  if (var < 0.0) var = 0.0;
  const float m = (float)mean, sd = (float)sqrt(var);
  adv[i] = (adv_raw[i] - m) / (sd + 1e-8f);
}

// keyed bijection on [0, B): 4-round Feistel on the enclosing power of four, cycle-walking
__global__ void hx_perm_kernel(int* perm, int B, uint32_t key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  int bits = 2;
  while ((1 << bits) < B) bits += 2;
  const int half = bits / 2;
  const uint32_t mask = (1u << half) - 1u;
  uint32_t x = (uint32_t)i;
  do {
    uint32_t l = x >> half, r = x & mask;
    for (int round = 0; round < 4; ++round) {
      uint32_t f = (r * 0x9E3779B1u) ^ (key + 0x85EBCA6Bu * (uint32_t)(round + 1));
      f ^= f >> 15; f *= 0x2C1B3C6Du; f ^= f >> 12;
      const uint32_t nl = r, nr = (l ^ f) & mask;
      l = nl; r = nr;
    }
    x = (l << half) | r;
  } while (x >= (uint32_t)B);
  perm[i] = (int)x;
}

// minibatch gather (rollout_storage.py:167-180): rows of obs / priv and the per-row scalars
struct GatherArgs {
  const int* idx; int M;
  const float* obs; int obs_ld; float* obs_mb;
  const float* priv; int priv_ld; float* priv_mb;
  const float* actions; const float* mu; const float* values; const float* returns; const float* logp; const float* adv;
  int A;
  float* row_mb;   // [M][2A+4]: actions A, mu_old A, value_old, return, logp_old, advantage
};
__global__ void __launch_bounds__(256) hx_gather_kernel(GatherArgs g) {
  const int m = blockIdx.x;
  const int src = g.idx[m];
  const f32x4* so = reinterpret_cast<const f32x4*>(g.obs + (size_t)src * g.obs_ld);
  f32x4* dobs = reinterpret_cast<f32x4*>(g.obs_mb + (size_t)m * g.obs_ld);
  for (int k = threadIdx.x; k < g.obs_ld / 4; k += blockDim.x) dobs[k] = so[k];
  const f32x4* sp = reinterpret_cast<const f32x4*>(g.priv + (size_t)src * g.priv_ld);
  f32x4* dp = reinterpret_cast<f32x4*>(g.priv_mb + (size_t)m * g.priv_ld);
  for (int k = threadIdx.x; k < g.priv_ld / 4; k += blockDim.x) dp[k] = sp[k];
  const int W = 2 * g.A + 4;
  float* r = g.row_mb + (size_t)m * W;
  if (threadIdx.x < g.A) { r[threadIdx.x] = g.actions[(size_t)src * g.A + threadIdx.x]; r[g.A + threadIdx.x] = g.mu[(size_t)src * g.A + threadIdx.x]; }
  if (threadIdx.x == 32) { r[2 * g.A] = g.values[src]; r[2 * g.A + 1] = g.returns[src]; r[2 * g.A + 2] = g.logp[src]; r[2 * g.A + 3] = g.adv[src]; }
}

// Single-frame storage: what the gather becomes.  For every position i of the permutation the row start / first valid element
// of both streams and the per-row scalars, in minibatch order -- 4 ints + 24 floats per row instead of 1668 floats.
struct MbTableArgs {
  const int* perm; int TN;
  const int* off_obs; const int* off_priv; const int* kz_obs; const int* kz_priv;
  int* mb_off_obs; int* mb_off_priv; int* mb_kz_obs; int* mb_kz_priv;
  const float* actions; const float* mu; const float* values; const float* returns; const float* logp; const float* adv;
  int A; float* row_mb;
};
// one thread per OUTPUT element (2A + 4 scalars + 4 table entries per row): coalesced stores, the scattered side is the 4-byte loads
__global__ void __launch_bounds__(256) hx_mb_tables_kernel(MbTableArgs g) {
  const int W = 2 * g.A + 8;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(idx / W), j = (int)(idx % W);
  if (i >= g.TN) return;
  const int src = g.perm[i];
  const int A = g.A, nf = 2 * A + 4;
  if (j < nf) {
    float v;
    if (j < A) v = g.actions[(size_t)src * A + j];
    else if (j < 2 * A) v = g.mu[(size_t)src * A + j - A];
    else if (j == 2 * A) v = g.values[src];
    else if (j == 2 * A + 1) v = g.returns[src];
    else if (j == 2 * A + 2) v = g.logp[src];
    else v = g.adv[src];
    g.row_mb[(size_t)i * nf + j] = v;
  } else if (j == nf) g.mb_off_obs[i] = g.off_obs[src];
  else if (j == nf + 1) g.mb_off_priv[i] = g.off_priv[src];
  else if (j == nf + 2) g.mb_kz_obs[i] = g.kz_obs[src];
  else g.mb_kz_priv[i] = g.kz_priv[src];
}
// rows [count][ld] of a frame-stored stream, expanded (hx_ppo_storage_rows; the deferred critic's batches): the workgroups walk the
// rows; a wave's lanes take consecutive elements (row starts are only float-aligned: 4-byte accesses, whole 256-byte segments per
// wave and instruction on both sides)
__global__ void __launch_bounds__(256) hx_expand_rows_kernel(const float* __restrict__ base, const int* __restrict__ off, const int* __restrict__ kz, int klim, int ld,
                                                             float* __restrict__ dst, int count) {
  for (int r = blockIdx.x; r < count; r += gridDim.x) {
    const float* src = base + off[r];
    const int z = kz[r];
    for (int k = threadIdx.x; k < ld; k += blockDim.x) dst[(size_t)r * ld + k] = (k >= z && k < klim) ? src[k] : 0.f;
  }
}

// One minibatch of rows of BOTH streams, expanded from the frame rings through the permutation-ordered tables of hx_mb_tables_kernel
// into ordinary matrices (the row starts are only float-aligned: 4-byte accesses, coalesced along the row; the loads hit the L2 -- the
// 15-frame windows of consecutive steps of a robot overlap in 14 frames).  One workgroup per row.  What the gather launch is to row storage;
// kept per minibatch across the epochs like the gathered rows (the reference reuses one permutation, rollout_storage.py:149).
struct ExpandMbArgs { const float* obs; const float* priv; const int* off_o; const int* kz_o; const int* off_p; const int* kz_p; int klim_o, klim_p, ld_o, ld_p; float* dst_o; float* dst_p; };
__global__ void __launch_bounds__(256) hx_expand_mb_kernel(ExpandMbArgs g) {
  const int r = blockIdx.x;
  const float* so = g.obs + g.off_o[r]; const float* sp = g.priv + g.off_p[r];
  const int zo = g.kz_o[r], zp = g.kz_p[r];
  const int no = g.ld_o >> 2, np = g.ld_p >> 2;
  for (int q = threadIdx.x; q < no + np; q += 256) {          // 16-byte stores; four scalar loads each (measured 157 us per minibatch against 199 us with 4-byte stores)
    const bool ob = q < no;
    const int k = 4 * (ob ? q : q - no);
    const float* src = ob ? so : sp;
    const int z = ob ? zo : zp, lim = ob ? g.klim_o : g.klim_p;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (k + j >= z && k + j < lim) ? src[k + j] : 0.f;
    *reinterpret_cast<f32x4*>((ob ? g.dst_o + (size_t)r * g.ld_o : g.dst_p + (size_t)r * g.ld_p) + k) = v;
  }
}

// Loss head (ppo.py:128-166 forward of the last layers, all loss terms, and their backward down to the
// pre-activation gradient of the third hidden layer).  64 rows per workgroup.
// Outputs: dZ3a[M][hw], dZ3c[M][hwc] and one partial slab per workgroup:
//   [A*hw dW4 | A db4 | hwc dW4c | 1 db4c | A dstd | kl_sum, value_loss_sum, surrogate_sum, entropy_sum]
// hw / hwc are the last hidden widths of the actor / critic (they differ for the sibling tasks).
#define HEAD_ROWS 32
struct HeadArgs {
  const float* h3a; const float* h3c; int hw, hwc;
  const float* W4; const float* b4; const float* W4c; const float* b4c; const float* stdp; const float* sigma_old;
  const float* row_mb; int M, A;
  float clip, vcoef, ecoef; int use_clipped_value_loss;
  float* dz3a; float* dz3c; float* slab; int slab_w;
  int stage_c;    // 1: the critic's rows are staged in LDS like the actor's (narrow critics; hector).  0: a wide critic
                  // (hector_full's 768) would need 130 KB and leave one workgroup per CU, so its rows are reduced to the
                  // value while they stream in and re-read from L2 for the two later uses
};
// dynamic LDS of the loss head: staged rows of the actor (and of the critic when stage_c), both head weights, per-row results
static inline size_t head_lds_bytes(int hw, int hwc, int A, bool stage_c) {
  return (size_t)(HEAD_ROWS * (hw + 8) + (stage_c ? HEAD_ROWS * (hwc + 8) : 0) + A * hw + hwc + HEAD_ROWS * (A + 1) +
                  HEAD_ROWS * (4 + A) + HEAD_ROWS) * sizeof(float);
}
template <int MA>   // register-array bound on the action count (16 for hector's 10, 32 otherwise): loops over MA are unrolled
__global__ void __launch_bounds__(256) hx_loss_head_kernel(HeadArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // staged rows are hw + 8 floats apart: 16-byte aligned for vector access, and the 8 rows x 8 k of a wave's dot-product step
  // (lane = (row, part), k = part + 8 i) fall on 2 x 32 distinct banks; with hw + 1 and a contiguous k block per lane the four
  // even parts of a row shared a bank (SQ_LDS_BANK_CONFLICT was 48 % of the kernel's LDS cycles)
  const int hw = g.hw, hwc = g.hwc, A = g.A, hp = hw + 8, hpc = hwc + 8;
  float* sHa = sm;                       // [rows][hw+1]
  const bool stage_c = g.stage_c != 0;
  float* sHc = sHa + HEAD_ROWS * hp;     // [rows][hwc+1]   (empty when the critic is not staged)
  float* sW = sHc + (stage_c ? HEAD_ROWS * hpc : 0);     // [A][hw]
  float* sWc = sW + A * hw;              // [hwc]
  float* sD = sWc + hwc;                 // [rows][A+1]  dmu, dv
  float* sL = sD + HEAD_ROWS * (A + 1);  // [rows][4+A]  kl, vloss, sloss, entropy, dsigma[A]
  float* sV = sL + HEAD_ROWS * (4 + A);  // [rows]  value dot product of a critic that is not staged
  const int r0 = blockIdx.x * HEAD_ROWS;
  const int tid = threadIdx.x;
  // half a wave per row, 16 bytes per lane (widths are multiples of 64 floats): coalesced, no integer divisions
  const int wave = tid >> 6, lane = tid & 63, hmax = hw > hwc ? hw : hwc;
  for (int r = 2 * wave + (lane >> 5); r < HEAD_ROWS; r += 8) {
    const bool ok = (r0 + r) < g.M;
    const size_t grow = (size_t)(ok ? r0 + r : 0);
    float vp = 0.f;
    for (int k = 4 * (lane & 31); k < hmax; k += 128) {      // one loop for both nets: two loads in flight per trip
      if (k < hw) {
        f32x4 x = *reinterpret_cast<const f32x4*>(g.h3a + grow * hw + k);
        if (!ok) x = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(sHa + r * hp + k) = x;
      }
      if (k < hwc) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(g.h3c + grow * hwc + k);
        if (stage_c) {
          *reinterpret_cast<f32x4*>(sHc + r * hpc + k) = ok ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
          const f32x4 w = *reinterpret_cast<const f32x4*>(g.W4c + k);
#pragma unroll
          for (int q = 0; q < 4; ++q) vp = fmaf(x[q], w[q], vp);
        }
      }
    }
    if (!stage_c) {                                          // uniform: the 32 lanes of the row's half-wave sum their parts
      for (int o = 16; o > 0; o >>= 1) vp += __shfl_xor(vp, o);
      if ((lane & 31) == 0) sV[r] = ok ? vp : 0.f;
    }
  }
  for (int i = tid; i < A * hw; i += 256) sW[i] = g.W4[i];
  for (int i = tid; i < hwc; i += 256) sWc[i] = g.W4c[i];
  __syncthreads();
  {
    // 8 lanes per row: lane `part` owns k = part, part + 8, ... of both dot products, 3 xor-shuffles reduce them,
    // lane 0 of the row finishes the loss terms
    const int r = tid >> 3, part = tid & 7, m = r0 + r;
    float mu[MA];
#pragma unroll
    for (int j = 0; j < MA; ++j) mu[j] = 0.f;
    float v = 0.f;
    if (hw == hwc && stage_c) {           // one pass over k for both nets (hector: 128 / 128)
      for (int k = part; k < hw; k += 8) {
        const float x = sHa[r * hp + k];
        for (int j = 0; j < A; ++j) mu[j] = fmaf(x, sW[j * hw + k], mu[j]);
        v = fmaf(sHc[r * hpc + k], sWc[k], v);
      }
    } else {
      for (int k = part; k < hw; k += 8) {
        const float x = sHa[r * hp + k];
        for (int j = 0; j < A; ++j) mu[j] = fmaf(x, sW[j * hw + k], mu[j]);
      }
      if (stage_c) { for (int k = part; k < hwc; k += 8) v = fmaf(sHc[r * hpc + k], sWc[k], v); }
      else v = (part == 0) ? sV[r] : 0.f;
    }
    for (int o = 4; o > 0; o >>= 1) {
      for (int j = 0; j < A; ++j) mu[j] += __shfl_xor(mu[j], o);
      v += __shfl_xor(v, o);
    }
    // every lane of the row holds all the sums now; the loss terms of the row's actions are spread over its 8 lanes
    // (lane p takes actions p, p + 8, ...) and reduced with three more shuffles
    float* L = sL + r * (4 + A);
    float* D = sD + r * (A + 1);
    if (m < g.M) {                                   // uniform over the 8 lanes of a row
      const float invM = 1.0f / (float)g.M;
      const float* row = g.row_mb + (size_t)m * (2 * A + 4);
      v += g.b4c[0];
      float logp = 0.f, ent = 0.f, kl = 0.f;
      float dq[MA / 8], sq[MA / 8];
#pragma unroll
      for (int q = 0; q < MA / 8; ++q) {
        const int j = part + 8 * q;
        dq[q] = 0.f; sq[q] = 1.f;
        if (j < A) {
          float mj = 0.f;
#pragma unroll
          for (int jj = 0; jj < MA; ++jj) mj = (jj == j) ? mu[jj] : mj;      // register array, lane-dependent index
          mj += g.b4[j];
          const float sgj = mj * 0.f + g.stdp[j];
          const float d = row[j] - mj;
          const float ls = logf(sgj);
          logp += -(d * d) / (2.0f * sgj * sgj) - ls - LOG_SQRT_2PI;
          ent += 0.5f + LOG_SQRT_2PI + ls;
          const float so = g.sigma_old[j], dm = row[A + j] - mj;
          kl += logf(sgj / so + 1.e-5f) + (so * so + dm * dm) / (2.0f * sgj * sgj) - 0.5f;
          dq[q] = d; sq[q] = sgj;
        }
      }
      for (int o = 4; o > 0; o >>= 1) { logp += __shfl_xor(logp, o); ent += __shfl_xor(ent, o); kl += __shfl_xor(kl, o); }
      const float v_old = row[2 * A], ret = row[2 * A + 1], logp_old = row[2 * A + 2], adv = row[2 * A + 3];
      const float ratio = expf(logp - logp_old);
      const float lo = 1.0f - g.clip, hi = 1.0f + g.clip;
      const float s = -adv * ratio, sc = -adv * fminf(fmaxf(ratio, lo), hi);
      const float inr = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
      const float w = (s > sc) ? 1.f : ((s == sc) ? 0.5f + 0.5f * inr : inr);
      const float dlogp = -adv * w * ratio * invM;
#pragma unroll
      for (int q = 0; q < MA / 8; ++q) {
        const int j = part + 8 * q;
        if (j < A) {
          const float d = dq[q], sgj = sq[q];
          D[j] = dlogp * d / (sgj * sgj);
          L[4 + j] = dlogp * (d * d / (sgj * sgj * sgj) - 1.0f / sgj) - g.ecoef * invM / sgj;
        }
      }
      if (part == 0) {
        float vl, dv;
        if (g.use_clipped_value_loss) {
          const float vc = v_old + fminf(fmaxf(v - v_old, -g.clip), g.clip);
          const float la = (v - ret) * (v - ret), lb = (vc - ret) * (vc - ret);
          vl = fmaxf(la, lb);
          const float inv = (fabsf(v - v_old) <= g.clip) ? 1.f : 0.f;
          const float ga = 2.0f * (v - ret), gb = 2.0f * (vc - ret) * inv;
          dv = (la > lb) ? ga : ((la == lb) ? 0.5f * ga + 0.5f * gb : gb);
        } else {
          vl = (ret - v) * (ret - v);
          dv = 2.0f * (v - ret);
        }
        D[A] = g.vcoef * dv * invM;
        L[0] = kl; L[1] = vl; L[2] = fmaxf(s, sc); L[3] = ent;
      }
    } else if (part == 0) {
      for (int j = 0; j <= A; ++j) D[j] = 0.f;
      for (int j = 0; j < 4 + A; ++j) L[j] = 0.f;
    }
  }
  __syncthreads();
  // dZ3 = (W4^T dmu) * elu'(h3)   and   dZ3c = dv w4c * elu'(h3c)
  for (int r = 2 * wave + (lane >> 5); r < HEAD_ROWS; r += 8) {       // half a wave per row, four k per lane, 16-byte stores
    if (r0 + r >= g.M) continue;
    const float dv = sD[r * (A + 1) + A];
    for (int k = 4 * (lane & 31); k < hmax; k += 128) {
      if (k < hw) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < A; ++j) {
          const float dj = sD[r * (A + 1) + j];
          const f32x4 w = *reinterpret_cast<const f32x4*>(sW + j * hw + k);
#pragma unroll
          for (int q = 0; q < 4; ++q) s[q] = fmaf(dj, w[q], s[q]);
        }
        f32x4 o;
        const f32x4 ha4 = *reinterpret_cast<const f32x4*>(sHa + r * hp + k);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float ha = ha4[q]; o[q] = s[q] * (ha > 0.f ? 1.f : ha + 1.f); }
        *reinterpret_cast<f32x4*>(g.dz3a + (size_t)(r0 + r) * hw + k) = o;
      }
      if (k < hwc) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sWc + k);
        f32x4 o;
        f32x4 hc4;
        if (stage_c) {
          hc4 = *reinterpret_cast<const f32x4*>(sHc + r * hpc + k);
        } else {
          hc4 = *reinterpret_cast<const f32x4*>(g.h3c + (size_t)(r0 + r) * hwc + k);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float hc = hc4[q]; o[q] = dv * w[q] * (hc > 0.f ? 1.f : hc + 1.f); }
        *reinterpret_cast<f32x4*>(g.dz3c + (size_t)(r0 + r) * hwc + k) = o;
      }
    }
  }
  // partial parameter gradients of the two heads, summed over this workgroup's rows
  float* slab = g.slab + (size_t)blockIdx.x * g.slab_w;
  for (int i = tid; i < A * hw; i += 256) {
    const int j = i / hw, k = i % hw;
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s = fmaf(sD[r * (A + 1) + j], sHa[r * hp + k], s);
    slab[i] = s;
  }
  for (int k = tid; k < hwc; k += 256) {
    float s = 0.f;
    if (stage_c) {
      for (int r = 0; r < HEAD_ROWS; ++r) s = fmaf(sD[r * (A + 1) + A], sHc[r * hpc + k], s);
    } else {
      const int rows = (g.M - r0) < HEAD_ROWS ? (g.M - r0) : HEAD_ROWS;      // rows past M carry dv = 0 and are not read
      for (int r = 0; r < rows; ++r) s = fmaf(sD[r * (A + 1) + A], g.h3c[(size_t)(r0 + r) * hwc + k], s);
    }
    slab[A * hw + A + k] = s;
  }
  if (tid < A) {
    float s = 0.f, d = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) { s += sD[r * (A + 1) + tid]; d += sL[r * (4 + A) + 4 + tid]; }
    slab[A * hw + tid] = s;                      // db4
    slab[A * hw + A + hwc + 1 + tid] = d;        // dstd
  }
  if (tid == 32) {
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s += sD[r * (A + 1) + A];
    slab[A * hw + A + hwc] = s;                  // db4c
  }
  if (tid >= 64 && tid < 68) {
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s += sL[r * (4 + A) + (tid - 64)];
    slab[A * hw + A + hwc + 1 + A + (tid - 64)] = s;
  }
}

// stage 1 of the head-slab reduction: out[c][i] = sum of slabs [c*chunk, (c+1)*chunk) -- many workgroups
__global__ void hx_slab_chunk_kernel(const float* __restrict__ slab, int S, int slab_w, int chunk, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= slab_w) return;
  const int lo = c * chunk, hi = min(S, lo + chunk);
  float s = 0.f;
  for (int k = lo; k < hi; ++k) s += slab[(size_t)k * slab_w + i];
  out[(size_t)c * slab_w + i] = s;
}

// All split-K partial slabs of one minibatch (6 weight + 6 bias segments) summed in ONE launch: segment table in
// kernel arguments, each workgroup owns 256 consecutive elements of one segment.  Fixed summation order -> bitwise
// reproducible gradients.
// A segment may be a column range of its destination matrix: the slabs are compact [S][rows x cols], element i of a slab goes
// to dst[(i / cols) * ldd + i % cols]  (cols = ldd = count for a flat segment; cols is a multiple of 4).
#define HX_MAX_SEG 24
struct ReduceTable {
  const float* src[HX_MAX_SEG]; float* dst[HX_MAX_SEG];
  unsigned count[HX_MAX_SEG]; int S[HX_MAX_SEG]; unsigned block0[HX_MAX_SEG + 1];
  unsigned cols[HX_MAX_SEG], ldd[HX_MAX_SEG];
  int nseg;
};
__global__ void __launch_bounds__(256) hx_reduce_all_kernel(ReduceTable t) {
  int seg = 0;
#pragma unroll
  for (int k = 1; k < HX_MAX_SEG; ++k) if (k < t.nseg && blockIdx.x >= t.block0[k]) seg = k;
  // four consecutive elements per thread (segment lengths are multiples of 4), four slabs in flight per step; the
  // additions stay in slab order, so the sums are bit-identical to a one-element-at-a-time loop
  const unsigned i = ((blockIdx.x - t.block0[seg]) * 256u + threadIdx.x) * 4u;
  if (i >= t.count[seg]) return;
  const float* src = t.src[seg] + i;
  const size_t stride = t.count[seg];
  const int S = t.S[seg];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= S; k += 4) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)k * stride);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 1) * stride);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 2) * stride);
    const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (size_t)(k + 3) * stride);
    s = s + v0; s = s + v1; s = s + v2; s = s + v3;
  }
  for (; k < S; ++k) s = s + *reinterpret_cast<const f32x4*>(src + (size_t)k * stride);
  const unsigned cols = t.cols[seg];
  const size_t di = (cols == t.count[seg]) ? (size_t)i : (size_t)(i / cols) * t.ldd[seg] + (i % cols);
  *reinterpret_cast<f32x4*>(t.dst[seg] + di) = s;
}

// The same loss head on the matrix cores (v_mfma_f32_16x16x4_f32), for heads of up to 16 actions over staged rows (the hector
// shape: 10 actions, 128 / 128 wide): the VALU version spends its time on LDS-fed dot products at the one-wave issue rate
// (2.7 k vector + 0.8 k LDS instructions per wave, half of a wave's life waiting; profiles/r03_h2).  Four products per 32 rows:
//   mu  [32 x 16]  = H3a [32 x hw] W4^T            waves 0, 1 (16 rows each), K = hw
//   v   [32 x 1]   = H3c [32 x hwc] w4c            waves 2, 3, K = hwc (B operand: w4c in column 0)
//   dZ3a[32 x hw]  = dmu [32 x 16] W4 [16 x hw]    all waves: one row tile x hw / 32 column tiles each, K = 16
//   dW4 [16 x hw]  = dmu^T [16 x 32] H3a [32 x hw] all waves: hw / 64 column tiles each, K = 32
// Operand layout of the instruction (as in the fused actor): lane l holds A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15];
// four MFMAs consume k = 4 (l >> 4) + i of a 16-deep block; C: col = l & 15, row = 4 (l >> 4) + reg.  Same outputs and the same
// slab layout as hx_loss_head_kernel; sums over k, actions and rows are associated differently (fp32 round-off).
#define HEADM_NA 16
static inline size_t head_mfma_lds_bytes(int hw, int hwc) {
  return (size_t)(HEAD_ROWS * (hw + 4) + HEAD_ROWS * (hwc + 4) + HEADM_NA * (hw + 4) + hwc + HEAD_ROWS * HEADM_NA * 2 + HEAD_ROWS + HEAD_ROWS * 4) * sizeof(float);
}
__global__ void __launch_bounds__(256) hx_loss_head_mfma_kernel(HeadArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int hw = g.hw, hwc = g.hwc, A = g.A, lda = hw + 4, ldc = hwc + 4;
  float* sHa = sm;                              // [32][hw + 4]
  float* sHc = sHa + HEAD_ROWS * lda;           // [32][hwc + 4]
  float* sW = sHc + HEAD_ROWS * ldc;            // [16][hw + 4]   W4, rows >= A zero
  float* sWc = sW + HEADM_NA * lda;             // [hwc]
  float* sD = sWc + hwc;                        // [32][16]  dmu (0 for actions >= A and rows >= M)
  float* sS = sD + HEAD_ROWS * HEADM_NA;        // [32][16]  dsigma terms
  float* sDv = sS + HEAD_ROWS * HEADM_NA;       // [32]      dv
  float* sL = sDv + HEAD_ROWS;                  // [32][4]   kl, value loss, surrogate, entropy
  const int r0 = blockIdx.x * HEAD_ROWS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, q4 = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- stage the rows (half a wave per row, 16 bytes per lane) and the two head weights
  {
    const int hmax = hw > hwc ? hw : hwc;
    for (int r = 2 * wave + (lane >> 5); r < HEAD_ROWS; r += 8) {
      const bool ok = (r0 + r) < g.M;
      const size_t grow = (size_t)(ok ? r0 + r : 0);
      for (int k = 4 * (lane & 31); k < hmax; k += 128) {
        if (k < hw) { const f32x4 x = *reinterpret_cast<const f32x4*>(g.h3a + grow * hw + k); *reinterpret_cast<f32x4*>(sHa + r * lda + k) = ok ? x : zero4; }
        if (k < hwc) { const f32x4 x = *reinterpret_cast<const f32x4*>(g.h3c + grow * hwc + k); *reinterpret_cast<f32x4*>(sHc + r * ldc + k) = ok ? x : zero4; }
      }
    }
    for (int i = tid; i < HEADM_NA * (hw >> 2); i += 256) {
      const int j = i / (hw >> 2), k = (i % (hw >> 2)) << 2;
      *reinterpret_cast<f32x4*>(sW + j * lda + k) = (j < A) ? *reinterpret_cast<const f32x4*>(g.W4 + (size_t)j * hw + k) : zero4;
    }
    for (int i = tid; i < hwc; i += 256) sWc[i] = g.W4c[i];
  }
  __syncthreads();
  // ---- forward heads + loss terms
  if (wave < 2) {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = sHa + (16 * wave + l15) * lda + 4 * q4;
    const float* bp = sW + l15 * lda + 4 * q4;
    for (int kb = 0; kb < hw; kb += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kb), b = *reinterpret_cast<const f32x4*>(bp + kb);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
    }
    const int j = l15;
    const bool jv = j < A;
    const float b4j = jv ? g.b4[j] : 0.f, sdj = jv ? g.stdp[j] : 1.f, soj = jv ? g.sigma_old[j] : 1.f;
    const float invM = 1.0f / (float)g.M;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = 16 * wave + 4 * q4 + reg, m = r0 + r;
      const bool rv = m < g.M;                                   // uniform over the 16 lanes of the row
      const float* row = g.row_mb + (size_t)(rv ? m : 0) * (2 * A + 4);
      float lp = 0.f, en = 0.f, kl = 0.f, d = 0.f, sgj = 1.f;
      if (jv) {
        const float mj = acc[reg] + b4j;
        sgj = mj * 0.f + sdj;
        d = row[j] - mj;
        const float ls = logf(sgj);
        lp = -(d * d) / (2.0f * sgj * sgj) - ls - LOG_SQRT_2PI;
        en = 0.5f + LOG_SQRT_2PI + ls;
        const float dm = row[A + j] - mj;
        kl = logf(sgj / soj + 1.e-5f) + (soj * soj + dm * dm) / (2.0f * sgj * sgj) - 0.5f;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); en += __shfl_xor(en, o); kl += __shfl_xor(kl, o); }
      const float logp_old = row[2 * A + 2], adv = row[2 * A + 3];
      const float ratio = expf(lp - logp_old);
      const float lo = 1.0f - g.clip, hi = 1.0f + g.clip;
      const float s = -adv * ratio, sc = -adv * fminf(fmaxf(ratio, lo), hi);
      const float inr = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
      const float w = (s > sc) ? 1.f : ((s == sc) ? 0.5f + 0.5f * inr : inr);
      const float dlogp = -adv * w * ratio * invM;
      const bool on = rv && jv;
      sD[r * HEADM_NA + j] = on ? dlogp * d / (sgj * sgj) : 0.f;
      sS[r * HEADM_NA + j] = on ? dlogp * (d * d / (sgj * sgj * sgj) - 1.0f / sgj) - g.ecoef * invM / sgj : 0.f;
      if (j == 0) { sL[r * 4 + 0] = rv ? kl : 0.f; sL[r * 4 + 2] = rv ? fmaxf(s, sc) : 0.f; sL[r * 4 + 3] = rv ? en : 0.f; }
    }
  } else {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    const int rt = wave - 2;
    const float* ap = sHc + (16 * rt + l15) * ldc + 4 * q4;
    const float* bp = sWc + 4 * q4;
    for (int kb = 0; kb < hwc; kb += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kb);
      f32x4 b = *reinterpret_cast<const f32x4*>(bp + kb);
      if (l15 != 0) b = zero4;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
    }
    if (l15 == 0) {
      const float invM = 1.0f / (float)g.M, b4c = g.b4c[0];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * rt + 4 * q4 + reg, m = r0 + r;
        float vl = 0.f, dvv = 0.f;
        if (m < g.M) {
          const float* row = g.row_mb + (size_t)m * (2 * A + 4);
          const float v = acc[reg] + b4c, v_old = row[2 * A], ret = row[2 * A + 1];
          float dv;
          if (g.use_clipped_value_loss) {
            const float vc = v_old + fminf(fmaxf(v - v_old, -g.clip), g.clip);
            const float la = (v - ret) * (v - ret), lb = (vc - ret) * (vc - ret);
            vl = fmaxf(la, lb);
            const float inv = (fabsf(v - v_old) <= g.clip) ? 1.f : 0.f;
            const float ga = 2.0f * (v - ret), gb = 2.0f * (vc - ret) * inv;
            dv = (la > lb) ? ga : ((la == lb) ? 0.5f * ga + 0.5f * gb : gb);
          } else {
            vl = (ret - v) * (ret - v);
            dv = 2.0f * (v - ret);
          }
          dvv = g.vcoef * dv * invM;
        }
        sDv[r] = dvv; sL[r * 4 + 1] = vl;
      }
    }
  }
  __syncthreads();
  // ---- dZ3a = (dmu W4) * elu'(h3a): wave = (row tile, half of the column tiles)
  {
    const int rt = wave & 1, nct = hw >> 4, c0 = (wave >> 1) * (nct >> 1);
    const f32x4 a = *reinterpret_cast<const f32x4*>(sD + (16 * rt + l15) * HEADM_NA + 4 * q4);
    for (int ct = c0; ct < c0 + (nct >> 1); ++ct) {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      const int col = 16 * ct + l15;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], sW[(4 * q4 + i) * lda + col], acc, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * rt + 4 * q4 + reg;
        if (r0 + r < g.M) { const float ha = sHa[r * lda + col]; g.dz3a[(size_t)(r0 + r) * hw + col] = acc[reg] * (ha > 0.f ? 1.f : ha + 1.f); }
      }
    }
  }
  // ---- dZ3c = dv w4c * elu'(h3c): half a wave per row, 16-byte stores
  for (int r = 2 * wave + (lane >> 5); r < HEAD_ROWS; r += 8) {
    if (r0 + r >= g.M) continue;
    const float dv = sDv[r];
    for (int k = 4 * (lane & 31); k < hwc; k += 128) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(sWc + k), hc4 = *reinterpret_cast<const f32x4*>(sHc + r * ldc + k);
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = dv * w[q] * (hc4[q] > 0.f ? 1.f : hc4[q] + 1.f);
      *reinterpret_cast<f32x4*>(g.dz3c + (size_t)(r0 + r) * hwc + k) = o;
    }
  }
  // ---- partial parameter gradients of the two heads, summed over this workgroup's rows
  float* slab = g.slab + (size_t)blockIdx.x * g.slab_w;
  {
    const int nct = hw >> 4, per = nct >> 2;              // dW4 = dmu^T H3a: column tiles split over the four waves
    for (int ct = wave * per; ct < (wave + 1) * per; ++ct) {
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      const int col = 16 * ct + l15;
#pragma unroll
      for (int kb = 0; kb < HEAD_ROWS; kb += 16)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = kb + 4 * q4 + i;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[r * HEADM_NA + l15], sHa[r * lda + col], acc, 0, 0, 0);
        }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) { const int j = 4 * q4 + reg; if (j < A) slab[j * hw + col] = acc[reg]; }
    }
  }
  for (int k = tid; k < hwc; k += 256) {
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s = fmaf(sDv[r], sHc[r * ldc + k], s);
    slab[A * hw + A + k] = s;
  }
  if (tid < A) {
    float s = 0.f, d = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) { s += sD[r * HEADM_NA + tid]; d += sS[r * HEADM_NA + tid]; }
    slab[A * hw + tid] = s;                      // db4
    slab[A * hw + A + hwc + 1 + tid] = d;        // dstd
  }
  if (tid == 32) {
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s += sDv[r];
    slab[A * hw + A + hwc] = s;                  // db4c
  }
  if (tid >= 64 && tid < 68) {
    float s = 0.f;
    for (int r = 0; r < HEAD_ROWS; ++r) s += sL[r * 4 + (tid - 64)];
    slab[A * hw + A + hwc + 1 + A + (tid - 64)] = s;
  }
}

// scatter the head slab sums into the flat gradient buffer + statistics
struct HeadScatter { size_t w4, b4, w4c, b4c, stdo, stats; int A, hw, hwc; };
__global__ void hx_head_scatter_kernel(const float* __restrict__ slab, int S, int slab_w, float* grads, HeadScatter o, float rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slab_w) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += slab[(size_t)k * slab_w + i];
  const int A = o.A, hw = o.hw, hwc = o.hwc;
  size_t dst;
  if (i < A * hw) dst = o.w4 + i;
  else if (i < A * hw + A) dst = o.b4 + (i - A * hw);
  else if (i < A * hw + A + hwc) dst = o.w4c + (i - A * hw - A);
  else if (i < A * hw + A + hwc + 1) dst = o.b4c;
  else if (i < A * hw + A + hwc + 1 + A) dst = o.stdo + (i - (A * hw + A + hwc + 1));
  else {
    const int q = i - (A * hw + A + hwc + 1 + A);   // 0 kl, 1 vloss, 2 sloss, 3 entropy
    if (q == 3) return;
    dst = o.stats + q;
  }
  grads[dst] = s;
  if (i == 0) grads[o.stats + 3] = rows;
}

// sum of squares of the gradient (clip_grad_norm_, ppo.py:173)
__global__ void __launch_bounds__(256) hx_sumsq_kernel(const float* __restrict__ g, size_t count, float scale, double* out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const double x = (double)(g[i] * scale);
    s += x * x;
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(out, sh[0]);
}

// adaptive-KL learning-rate schedule (ppo.py:136-148) + running loss sums, on device so no host sync is needed
struct SchedState { float lr; float last_kl; float vloss_sum; float sloss_sum; };
__global__ void hx_schedule_kernel(const float* __restrict__ stats, int adaptive, float desired_kl, SchedState* st, double* sumsq) {
  *sumsq = 0.0;                          // accumulator of the gradient-norm kernel that follows (was a memset launch per optimiser step)
  const float rows = stats[3];
  const float kl = stats[0] / rows;
  float lr = st->lr;
  if (adaptive) {
    if (kl > desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
    else if (kl < desired_kl / 2.0f && kl > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
  }
  st->lr = lr;
  st->last_kl = kl;
  st->vloss_sum += stats[1] / rows;
  st->sloss_sum += stats[2] / rows;
}

// clip_grad_norm_ + Adam (ppo.py:171-174 ; torch.optim.Adam defaults betas (0.9,0.999) eps 1e-8)
__global__ void __launch_bounds__(256) hx_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, size_t count, float gscale, const double* sumsq,
                                                      float max_norm, const SchedState* st, float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float total = (float)sqrt(*sumsq);
  if (!(total < 3.0e38f)) return;       // non-finite gradient (a NaN reward or observation upstream): skip the step, keep weights and moments
  const float coef = fminf(1.0f, max_norm / (total + 1e-6f));
  const float gi = g[i] * gscale * coef;
  const float mi = 0.9f * m[i] + (1.0f - 0.9f) * gi;
  const float vi = 0.999f * v[i] + (1.0f - 0.999f) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + 1e-8f;
  p[i] = p[i] - (st->lr / bc1) * (mi / denom);
}

// ================================================================= host side
struct Layer { int out, in, in_ld; size_t w, b; };

// Rollout slots per deferred critic batch.  Measured at 4096 envs (profiles/r02_e_critic_chunk.txt): a batch of 2 slots
// (8192 rows, 128 x 128 BK16 tiles) starting beside the actor kernel costs the rollout 1.0 ms per iteration, 5 slots 2.7 ms,
// 1 slot 2.1-3.8 ms; starting the batch after the actor (beside the env step only) is worse for every size.
#ifndef HX_CRITIC_CHUNK
#define HX_CRITIC_CHUNK 2
#endif
struct hx_ppo {
  hx_ppo_cfg cfg;
  hipStream_t stream; bool own_stream;
  hipStream_t stream2; hipEvent_t ev_fork, ev_join;   // critic chain of the rollout runs beside the actor chain
  hipStream_t stream_b = nullptr; hipEvent_t ev_b0 = nullptr, ev_b1 = nullptr;   // update phase, optional (HX_UPDATE_STREAMS=2): the critic's GEMM chain beside the actor's
  Layer L[8];                  // actor 0..3, critic 4..7
  size_t std_off, padded, stats_off;
  int64_t torch_count;
  float *params, *grads, *m, *v; bool ext_grads;
  // storage
  float *s_obs, *s_priv, *s_actions, *s_mu, *s_values, *s_logp, *s_rewards, *s_returns, *s_adv_raw, *s_adv, *sigma_old;
  unsigned char* s_dones; unsigned char* s_timeouts;
  // single-frame observation storage (hx_ppo_cfg.obs_frame ...; include/hx_sim.h): s_obs / s_priv are per-robot frame rings
  // [N][Po][fo] / [N][Pp][fp] then; row (t, e) starts at off_*[t * N + e] and its first kz_*[t * N + e] elements read as zero
  bool frames = false;
  int fo = 0, fp = 0, So = 0, Sp = 0, Po = 0, Pp = 0;
  int *off_obs = nullptr, *off_priv = nullptr;          // [(T + 1) * N] row starts (float offsets), fixed at creation
  int *kz_obs = nullptr, *kz_priv = nullptr;            // [(T + 1) * N] first valid element of every stored row, written by the env step
  int *mb_off_obs = nullptr, *mb_off_priv = nullptr, *mb_kz_obs = nullptr, *mb_kz_priv = nullptr;   // [T * N] the same in minibatch (permutation) order
  long long frames_sim_step = -1;                        // simulator step counter right after the last frame-mode rollout step
  int crit_done;                 // rollout slots [0, crit_done) already have their critic values
  hipEvent_t ev_priv, ev_crit;   // priv rows of a slot copied (main stream) / deferred critic finished (stream2)
  int critic_chunk;              // rollout slots per deferred critic batch (HX_CRITIC_CHUNK, default 2)
  float* apack[3]; bool apack_dirty;   // actor hidden-layer weights in MFMA fragment order for the fused rollout actor; stale after any parameter change
  int critic_late;               // 1: a deferred critic batch starts after the actor kernel of its step instead of beside it
  int fwd_in_tile;               // rows per tile of the input layers' forward products at update size (HX_FWD_IN_TILE, 128 or 64)
  int wgrad_group;               // 1: the weight-gradient products of a minibatch go out as grouped split-K launches (HX_WGRAD_GROUP)
  int wgrad_multi;               // 1: ... at one workgroup per CU with per-product tile shapes (hx_wgrad_multi_kernel, HX_WGRAD_MULTI)
  std::map<int, WgradPlan> wplans;          // per minibatch row count
  float *wslab = nullptr, *wbslab = nullptr; size_t wslab_floats = 0, wbslab_floats = 0;
  int frames_gather;             // 1: frame storage feeds the first-layer products through gathered loaders (round 3); 0: rows expanded once per minibatch and update (HX_FRAMES_GATHER)
  int gemm_sp;                   // 1: long-K hidden-layer pairs of the update through the slot-placed loop (HX_GEMM_SP)
  int gemm_pair;                 // 1: layer l of the actor and of the critic share one launch in the update (HX_GEMM_PAIR)
  int head_mfma;                 // 1: hector-shaped loss heads run on the matrix cores (HX_HEAD_MFMA)
  int* pause_flag = nullptr;     // count of fused-actor workgroups in flight; the background critic sleeps while it is up (HX_CRITIC_YIELD)
  int bg_persist;                // > 0: the background critic's GEMMs run on this many persistent workgroups (HX_BG_PERSIST)
  int bg_waves;                  // 4: hx_gemm_persistent_kernel (four waves per workgroup, half the CUs); 2: hx_gemm_sp_persistent_kernel (two waves, every CU) (HX_BG_WAVES)
  int stack_pause;               // HX_STACK_PAUSE: the background critic sleeps through the simulator's stacking launch too
  int bg_tile;                   // experiment knob HX_BG_TILE: rows per tile of the background critic's GEMMs (0 = by batch size)
  float* last_values; double* moments;
  int step;
  // workspace
  int Mmax;
  float *obs_mb, *priv_mb, *row_mb;
  float *obs_mb_all, *priv_mb_all, *row_mb_all; int mb_slots; unsigned long long mb_gathered;   // gathered rows kept per minibatch
  float *act_a[3], *act_c[3], *dz_a[3], *dz_c[3];
  float *slab, *bias_slab, *head_slab, *head_slab2; size_t slab_floats;
  size_t slab_off[8], bslab_off[8];     // per-layer regions so that all six wgrads finish before one reduce launch
  int slab_splits[8];                   // split-K partial slabs allocated per layer: gemm_wgrad never writes more
  int head_blocks_max, head_slab_w;
  int* perm; int perm_external;
  double* sumsq; SchedState* sched;
  int64_t adam_t;
  int mb_done, mb_total;
  int prof_only;                 // -1: every symbol's launches are bracketed with HIP events while profiling; else only this registry id
  int prof_every = 1; long prof_seen = 0;      // bracket every prof_every-th launch of the selected symbols
  bool bf16;                     // forward / dgrad products on the bf16 matrix cores (hx_gemm_bf16.h)
  int actor_waves;               // waves per workgroup of the fused rollout actor (8; 4 with HX_ACTOR_WAVES=4)
  int actor_depth;               // weight blocks in flight per wave of the fused actor: 2 (hand-written loop), 3, 4, 6 (HX_ACTOR_DEPTH)
  int actor_rows;                // rows per workgroup of the fused rollout actor: 0 = by batch size, 16, 32 (HX_ACTOR_ROWS)
  float* wT[8];                  // fp32 transposed copies W^T[in][out] of the hidden weights the dgrads need (bf16 mode)
  uint32_t seed_lo, seed_hi, act_counter, perm_counter, perm_key;
  uint32_t row_base = 0;         // global index of this learner's first env row (hx_ppo_set_row_base): counter word of the action-noise stream
  hx_comm* comm = nullptr;       // RCCL communicator (hx_comm.hip) when data parallel: all-reduce inside hx_ppo_minibatch_step
  // profiling
  bool prof; std::vector<hipEvent_t> ev; std::vector<int> ev_kid; size_t ev_used; std::vector<double> prof_flops; std::vector<long> prof_launches;
  std::vector<void*> allocs;
};

// every host wait on the learner's stream: with a communicator the stream may carry a collective whose peer is gone, so the wait
// has the communicator's deadline (hx_comm_wait: abort + error after HX_COMM_TIMEOUT_S), never an unbounded hipStreamSynchronize
static int learner_sync(hx_ppo* s) {
  if (s->comm) return hx_comm_wait(s->comm, s->stream, 0.0);
  HX_CHECK(hipStreamSynchronize(s->stream));
  return 0;
}

template <typename T> static int palloc(hx_ppo* s, T** ptr, size_t count) {
  HX_CHECK(hipMalloc((void**)ptr, count * sizeof(T)));
  HX_CHECK(hipMemsetAsync(*ptr, 0, count * sizeof(T), s->stream));
  s->allocs.push_back(*ptr);
  return 0;
}

// ---- launch profiler (include/hx_lab.h): ONE row per kernel symbol, named exactly as rocprofv3 prints the symbol (template
// arguments included), so that a HIP-event row of bench.py and a rocprofv3 --stats row are the same launches.
static std::vector<std::string>& prof_names() { static std::vector<std::string> v; return v; }
static int prof_register(const std::string& n) { prof_names().push_back(n); return (int)prof_names().size() - 1; }
static const char* tf(bool b) { return b ? "true" : "false"; }
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, bool KFULL, bool GA, bool GB> static int gemm_kid() {
  static const int id = prof_register("hx_gemm_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(BKT) + ", " + tf(AK) + ", " +
                                      tf(BK_) + ", " + std::to_string(EPI) + ", " + tf(KFULL) + ", " + tf(GA) + ", " + tf(GB) + ">");
  return id;
}
// brackets one launch with HIP events on its stream when the symbol is selected; returns true if it did
struct ProfScope {
  hx_ppo* s; hipStream_t st; bool on;
  ProfScope(hx_ppo* s_, int kid, hipStream_t st_, double flops);
  ~ProfScope();
};

// ---- GEMM dispatch
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, bool KFULL = false, bool GA = false, bool GB = false> static void launch_gemm(hx_ppo* s, GemmArgs& g, hipStream_t st) {
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  const int blocks = g.tiles_m * g.tiles_n * (EPI == EPI_SLAB ? g.splits : 1);
  ProfScope ps(s, gemm_kid<BM, BN, BKT, AK, BK_, EPI, KFULL, GA, GB>(), st, 2.0 * g.M * g.N * g.K);
  hipLaunchKernelGGL((hx_gemm_kernel<BM, BN, BKT, AK, BK_, EPI, KFULL, GA, GB>), dim3(blocks), dim3(256), 0, st, g);
}

// bf16-input variant (hx_gemm_bf16.h)
template <int EPI> static void launch_gemm_bf16(hx_ppo* s, GemmArgs& g, hipStream_t st) {
  g.tiles_m = (g.M + 127) / 128;
  g.tiles_n = (g.N + 127) / 128;
  const int blocks = g.tiles_m * g.tiles_n;
  static const int kid = prof_register("hx_gemm_bf16_kernel<" + std::to_string(EPI) + ">");
  ProfScope ps(s, kid, st, 2.0 * g.M * g.N * g.K);
  hipLaunchKernelGGL((hx_gemm_bf16_kernel<EPI>), dim3(blocks), dim3(256), 0, st, g);
}

// W[out][ld_in] (first `in` columns) -> WT[in][out]
__global__ void hx_transpose_kernel(const float* __restrict__ W, int out, int in, int ld_in, float* __restrict__ WT) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int o = by + r, i = bx + threadIdx.x;
    t[r][threadIdx.x] = (o < out && i < in) ? W[(size_t)o * ld_in + i] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int i = bx + r, o = by + threadIdx.x;
    if (i < in && o < out) WT[(size_t)i * out + o] = t[threadIdx.x][r];
  }
}
static void refresh_transposes(hx_ppo* s, hipStream_t st) {
  if (!s->bf16) return;
  for (int net = 0; net < 2; ++net)
    for (int l = 1; l <= 2; ++l) {
      const Layer& L = s->L[net * 4 + l];
      const int in = L.in_ld;      // hidden widths are multiples of 4: in_ld == fan-in
      hipLaunchKernelGGL(hx_transpose_kernel, dim3((in + 31) / 32, (L.out + 31) / 32), dim3(32, 8), 0, st, s->params + L.w, L.out, in, L.in_ld,
                         s->wT[net * 4 + l]);
    }
}

// Variant choice is from measurement (tools/gemm_bench.py, profiles/r01_gemm_variants*.txt): after the branch-free
// epilogue all variants are within ~5 %; BK = 32 is best for the forward layers and 64-row BK = 32 tiles for dgrad
// (short K = 128..256, where a shorter launch tail matters most).
// rows given as windows of the frame rings (first layer of a network with single-frame storage): table slices for the M rows
struct RowTable { const int* off; const int* kz; int klim; float* copy = nullptr; int copy_ld = 0; };
static void gemm_fwd(hx_ppo* s, hipStream_t st, const float* X, int ldx, const float* W, int ldw, const float* b, float* Y, int M, int N, int K,
                     bool background = false, bool fp32_only = false, const RowTable* rt = nullptr) {
  GemmArgs g{};
  g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b;
  if (background) g.pause = s->pause_flag;      // persistent launches only look at it
  if (rt) {
    // gathered A rows: BK16 tiles (K = 616 / 1052 are not multiples of 32), 128-row tiles at update size, 64-row below, the
    // persistent half-chip grid for the rollout's background critic
    g.a_off = rt->off; g.a_kz = rt->kz; g.a_klim = rt->klim; g.a_copy = rt->copy; g.a_copy_ld = rt->copy_ld;
    if (background && s->bg_persist > 0) {
      g.tiles_m = (g.M + 63) / 64; g.tiles_n = (g.N + 127) / 128;
      hipLaunchKernelGGL((hx_gemm_persistent_kernel<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU, false, true, false>), dim3(s->bg_persist), dim3(256), 0, st, g, g.tiles_m * g.tiles_n);
    }
    else if (M >= 8192) launch_gemm<128, 128, 16, true, true, EPI_BIAS_ELU, false, true, false>(s, g, st);
    else launch_gemm<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU, false, true, false>(s, g, st);
    return;
  }
  // K a multiple of 32: 128x128 tiles, BK 32.  The two input layers (K = 616 / 1052) would pad 24 / 4 k-steps per tile
  // at BK 32; 64-row BK 16 tiles waste less and measured 3-6 % faster there (profiles/r01_e_gemm_loops.txt).
  // Background launches (the deferred critic beside the rollout) take the 128x128 BK16 kernel: 37 KB of LDS per
  // workgroup leaves room for an env-step workgroup (34 KB) next to three of them on a CU; with the 74 KB BK32 tiles
  // the env-step kernel waited for LDS (440 us instead of 200 us on the steps a critic burst overlaps,
  // profiles/r01_j_rollout_interference.txt).
  if (s->bf16 && !fp32_only) launch_gemm_bf16<EPI_BIAS_ELU>(s, g, st);   // every hidden-layer forward product in bf16 mode
  else if (background && s->bg_persist > 0 && s->bg_waves == 2) {
    // two-wave workgroups of the slot-placed loop, one per CU (hx_gemm_sp_persistent_kernel): two SIMDs of every CU stay free
    g.tiles_m = (g.M + 63) / 64; g.tiles_n = (g.N + 127) / 128;
    if (K % 16 == 0) hipLaunchKernelGGL((hx_gemm_sp_persistent_kernel<64, 128, 16, true, true, EPI_BIAS_ELU, 1, 2, true>), dim3(s->bg_persist), dim3(128), 0, st, g, g.tiles_m * g.tiles_n);
    else hipLaunchKernelGGL((hx_gemm_sp_persistent_kernel<64, 128, 16, true, true, EPI_BIAS_ELU, 1, 2, false>), dim3(s->bg_persist), dim3(128), 0, st, g, g.tiles_m * g.tiles_n);
  }
  else if (background && s->bg_persist > 0) {
    // the rollout's background critic on a small fixed grid: see hx_gemm_persistent_kernel
    // 128-row tiles since round 4: with the env step 16 us shorter (hx_dyn.h shape_gap_points) the 64-row tiles that round 2 preferred
    // (profiles/r02_e) no longer finish a two-slot batch inside its two steps and the remainder lands in the update (+2.5 ms);
    // the 128-row tiles do (profiles/r04_ba_critic_tiles.txt).  HX_BG_TILE=64 restores the old choice.
    if (s->bg_tile == 64) {
      g.tiles_m = (g.M + 63) / 64; g.tiles_n = (g.N + 127) / 128;
      hipLaunchKernelGGL((hx_gemm_persistent_kernel<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU>), dim3(s->bg_persist), dim3(256), 0, st, g, g.tiles_m * g.tiles_n);
    } else {
      g.tiles_m = (g.M + 127) / 128; g.tiles_n = (g.N + 127) / 128;
      hipLaunchKernelGGL((hx_gemm_persistent_kernel<128, 128, 16, true, true, EPI_BIAS_ELU>), dim3(s->bg_persist), dim3(256), 0, st, g, g.tiles_m * g.tiles_n);
    }
  }
  else if (background && (s->bg_tile == 128 || (s->bg_tile == 0 && M >= 8192))) launch_gemm<128, 128, 16, true, true, EPI_BIAS_ELU>(s, g, st);
  else if (background && s->bg_tile == 64) launch_gemm<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU>(s, g, st);
  else if (M >= 16384 && K % 32 == 0) launch_gemm<128, 128, 32, true, true, EPI_BIAS_ELU, true>(s, g, st);     // whole K tiles only
  // the two input layers (K = 616 / 1052, not multiples of 32) at update size: 128-row BK16 tiles -- 2880 tiles on 768 slots
  // (3.75 rounds) against 5760 on 1280 (4.5 rounds) for the critic's; re-measured in round 2: update 31.9 -> 31.6 ms
  // (profiles/r02_l_fwd_input_tiles.txt; HX_FWD_IN_TILE=64 restores the round-1 choice)
  else if (M >= 16384 && s->fwd_in_tile == 128) launch_gemm<128, 128, 16, true, true, EPI_BIAS_ELU>(s, g, st);
  else launch_gemm<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU>(s, g, st);
}
static void gemm_dgrad(hx_ppo* s, hipStream_t st, const float* dZ, int ldz, const float* W, int ldw, const float* H, float* dX, int M, int N, int K,
                       const float* WT = nullptr) {
  GemmArgs g{};
  g.A = dZ; g.lda = ldz; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = N; g.M = M; g.N = N; g.K = K; g.H = H; g.ldh = N;
  if (s->bf16 && WT != nullptr) { g.B = WT; g.ldb = K; launch_gemm_bf16<EPI_ELU_GRAD>(s, g, st); return; }   // B = W^T[N][K], K-major
  if (K % 32 == 0) launch_gemm<64, 128, 32, true, false, EPI_ELU_GRAD, true>(s, g, st);
  else launch_gemm<64, 128, 32, true, false, EPI_ELU_GRAD>(s, g, st);
}
// dW[out][in_ld] = dZ[Mrows][out]^T X[Mrows][in_ld] ; returns the number of splits written to slab; *bias_parts = number of
// partial rows written to bias_slab (splits x the tile_n blocks that share the column-sum work)
static int gemm_wgrad(hx_ppo* s, hipStream_t st, const float* dZ, int out, const float* X, int ldx, int in_ld, int Mrows, float* slab, float* bias_slab,
                      int* bias_parts, int alloc_splits) {
  GemmArgs g{};
  g.A = dZ; g.lda = out; g.B = X; g.ldb = ldx; g.C = slab; g.ldc = in_ld; g.M = out; g.N = in_ld; g.K = Mrows;
  const int tiles = ((out + 127) / 128) * ((in_ld + 127) / 128);
  // Split-K grid = ONE full wave of workgroups: 3 resident per CU (144 VGPRs -> 3 waves per SIMD; 32 KB LDS), never
  // more.  All workgroups of a split-K launch do the same work in lockstep, so a grid of 1026 blocks on 768 slots ran
  // a second, nearly empty round: the big layer went 1081 -> 901 us when the grid stopped exceeding the slot count
  // (profiles/r01_e_wgrad_blocks.txt).  HX_WGRAD_BLOCKS overrides the target for experiments.
  static int target_blocks = -1;
  if (target_blocks < 0) {
    const char* e = getenv("HX_WGRAD_BLOCKS");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    target_blocks = e ? atoi(e) : 3 * cus;
  }
  int splits = target_blocks / tiles;
  if (splits < 1) splits = 1;
  int max_splits = Mrows / HX_WGRAD_MIN_CHUNK; if (max_splits < 1) max_splits = 1;
  if (splits > max_splits) splits = max_splits;
  // never more partial slabs than ppo_create_impl allocated for this layer (a larger HX_WGRAD_BLOCKS override or a device
  // with more CUs would otherwise write past the layer's slab region); rounding the chunk up only lowers the count again
  if (splits > alloc_splits) splits = alloc_splits;
  int kchunk = rup((Mrows + splits - 1) / splits, 32);
  splits = (Mrows + kchunk - 1) / kchunk;
  if (s->bf16) {                        // K chunks in whole 64-deep tiles for the bf16 kernel
    kchunk = rup(kchunk, 64);
    splits = (Mrows + kchunk - 1) / kchunk;
  }
  g.splits = splits; g.kchunk = kchunk; g.dbias = bias_slab;
  g.db_parts = s->bf16 ? 1 : (in_ld + 127) / 128;
  *bias_parts = splits * g.db_parts;
  if (s->bf16) {
    g.tiles_m = (g.M + 127) / 128; g.tiles_n = (g.N + 127) / 128;
    const int blocks = g.tiles_m * g.tiles_n * g.splits;
    static const int kid = prof_register("hx_wgrad_bf16_kernel");
    ProfScope ps(s, kid, st, 2.0 * g.M * g.N * g.K);
    hipLaunchKernelGGL(hx_wgrad_bf16_kernel, dim3(blocks), dim3(256), 0, st, g);
    return splits;
  }
  // kchunk is a multiple of 32; with a row count that is a multiple of the K tile every split is whole tiles
  if (Mrows % HX_BK_UPD == 0) launch_gemm<128, 128, HX_BK_UPD, false, false, EPI_SLAB, true>(s, g, st);
  else launch_gemm<128, 128, HX_BK_UPD, false, false, EPI_SLAB>(s, g, st);
  return splits;
}

// ---- the weight-gradient products of one minibatch as grouped split-K launches (hx_gemm_group_kernel)
struct WgradJob { const float* dZ; int out; const float* X; int ldx, in_ld; float* slab; float* bias_slab; int alloc_splits; int seg; };
static int wgrad_target_blocks() {
  static int target_blocks = -1;
  if (target_blocks < 0) {
    const char* e = getenv("HX_WGRAD_BLOCKS");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    target_blocks = e ? atoi(e) : 3 * cus;
  }
  return target_blocks;
}
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, bool KFULL> static int gemm_group_kid() {
  static const int id = prof_register("hx_gemm_group_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(BKT) + ", " + tf(AK) + ", " +
                                      tf(BK_) + ", " + std::to_string(EPI) + ", " + tf(KFULL) + ">");
  return id;
}
// members filled except tiles_m / tiles_n / first[]
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, bool KFULL> static void launch_gemm_group(hx_ppo* s, GemmGroup& G, hipStream_t st) {
  int blocks = 0; double flops = 0.0;
  for (int i = 0; i < G.n; ++i) {
    GemmArgs& g = G.p[i];
    g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN;
    G.first[i] = blocks; blocks += g.tiles_m * g.tiles_n * (EPI == EPI_SLAB ? g.splits : 1);
    flops += 2.0 * g.M * g.N * g.K;
  }
  G.first[G.n] = blocks;
  ProfScope ps(s, gemm_group_kid<BM, BN, BKT, AK, BK_, EPI, KFULL>(), st, flops);
  hipLaunchKernelGGL((hx_gemm_group_kernel<BM, BN, BKT, AK, BK_, EPI, KFULL>), dim3(hx_group_grid(G)), dim3(256), 0, st, G);
}
// the same group through the slot-placed loop (hx_gemm_sp.h): bit-identical results; measured faster where the K loop is long enough
// for the loop to matter and the tile count fills the chip several times (profiles/r04_b_gemm_lab.txt: hidden-layer forward pair
// +1.7 %, first input-gradient pair +3.7 %; the 256 -> 128 products and the input layers' pair are not)
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, int WM, int WN, bool KFULL> static void launch_gemm_sp_group(hx_ppo* s, GemmGroup& G, hipStream_t st) {
  int blocks = 0; double flops = 0.0;
  for (int i = 0; i < G.n; ++i) {
    GemmArgs& g = G.p[i];
    g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN;
    G.first[i] = blocks; blocks += g.tiles_m * g.tiles_n * (EPI == EPI_SLAB ? g.splits : 1);
    flops += 2.0 * g.M * g.N * g.K;
  }
  G.first[G.n] = blocks;
  static const int kid = prof_register("hx_gemm_sp_group_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(BKT) + ", " + tf(AK) + ", " +
                                       tf(BK_) + ", " + std::to_string(EPI) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", " + tf(KFULL) + ">");
  ProfScope ps(s, kid, st, flops);
  hipLaunchKernelGGL((hx_gemm_sp_group_kernel<BM, BN, BKT, AK, BK_, EPI, WM, WN, KFULL>), dim3(hx_group_grid(G)), dim3(64 * WM * WN), 0, st, G);
}
// Cuts `jobs` into groups whose tiles fill one wave of workgroups (>= 95 % of the 3-per-CU slots at a whole number of
// slices per tile), peeling off the largest product while they do not, and launches each group.  splits_out[i] /
// bias_parts_out[i] = slices written to job i's slab / partial rows written to its bias slab.
static void gemm_wgrad_groups(hx_ppo* s, hipStream_t st, const WgradJob* jobs, int njobs, int Mrows, int* splits_out, int* bias_parts_out) {
  const int target = wgrad_target_blocks();
  int order[16], tiles[16];
  for (int i = 0; i < njobs; ++i) { order[i] = i; tiles[i] = ((jobs[i].out + 127) / 128) * ((jobs[i].in_ld + 127) / 128); }
  for (int i = 1; i < njobs; ++i)          // largest first (stable)
    for (int j = i; j > 0 && tiles[order[j]] > tiles[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  int pos = 0;
  while (pos < njobs) {
    int cnt = njobs - pos, T = 0;
    for (int i = pos; i < njobs; ++i) T += tiles[order[i]];
    while (cnt > 1) {          // the whole rest as one group, or without its largest members until it fits and fills
      const int S = target / T;
      if (cnt <= HX_GROUP_MAX && S >= 1 && (double)S * T >= 0.95 * target) break;
      // peel: the group becomes the single largest product
      cnt = 1; T = tiles[order[pos]];
    }
    int splits = target / T; if (splits < 1) splits = 1;
    int max_splits = Mrows / HX_WGRAD_MIN_CHUNK; if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    for (int i = pos; i < pos + cnt; ++i) if (splits > jobs[order[i]].alloc_splits) splits = jobs[order[i]].alloc_splits;
    const int kchunk = rup((Mrows + splits - 1) / splits, 32);
    splits = (Mrows + kchunk - 1) / kchunk;
    GemmGroup G{};
    G.n = cnt;
    for (int i = 0; i < cnt; ++i) {
      const WgradJob& j = jobs[order[pos + i]];
      GemmArgs& g = G.p[i];
      g.A = j.dZ; g.lda = j.out; g.B = j.X; g.ldb = j.ldx; g.C = j.slab; g.ldc = j.in_ld; g.M = j.out; g.N = j.in_ld; g.K = Mrows;
      g.splits = splits; g.kchunk = kchunk; g.dbias = j.bias_slab; g.db_parts = (j.in_ld + 127) / 128;
      splits_out[order[pos + i]] = splits; bias_parts_out[order[pos + i]] = splits * g.db_parts;
    }
    launch_gemm_group<128, 128, HX_BK_UPD, false, false, EPI_SLAB, true>(s, G, st);
    pos += cnt;
  }
}

// plan of the one-workgroup-per-CU weight gradients for minibatches of M rows (hx_wgrad_plan.h); slab buffers grow on demand
static int wgrad_plan_for(hx_ppo* s, const WgradLayerDesc* wl, int M, const WgradPlan** out) {
  auto it = s->wplans.find(M);
  if (it == s->wplans.end()) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    it = s->wplans.emplace(M, hx_wgrad_plan(wl, 6, M, cus)).first;
    if ((int)it->second.pieces.size() * 2 > HX_MAX_SEG) { hx_set_error("wgrad plan: too many pieces for the reduce table"); return -2; }
  }
  const WgradPlan& p = it->second;
  if (p.slab_floats > s->wslab_floats || p.bslab_floats > s->wbslab_floats) {
    if (int rc = learner_sync(s)) return rc;
    if (s->wslab) (void)hipFree(s->wslab);
    if (s->wbslab) (void)hipFree(s->wbslab);
    s->wslab = s->wbslab = nullptr; s->wslab_floats = s->wbslab_floats = 0;
    HX_CHECK(hipMalloc((void**)&s->wslab, p.slab_floats * sizeof(float))); s->wslab_floats = p.slab_floats;
    HX_CHECK(hipMalloc((void**)&s->wbslab, (p.bslab_floats + 4) * sizeof(float))); s->wbslab_floats = p.bslab_floats;
  }
  *out = &p;
  return 0;
}

extern "C" int hx_ppo_gemm_test(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* bias,
                                float* C, int ldc, const float* H, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.H = H; g.ldh = ldc;
  if (mode == 7) {                   // bf16 wgrad, single split, direct output; column sums of A -> `bias`
    g.splits = 1; g.kchunk = rup(K, 64); g.dbias = const_cast<float*>(bias);
    g.tiles_m = (M + 127) / 128; g.tiles_n = (N + 127) / 128;
    hipLaunchKernelGGL(hx_wgrad_bf16_kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, st, g);
    HX_CHECK(hipGetLastError());
    return 0;
  }
  if (mode == 5 || mode == 6) {      // bf16-input kernels: 5 = forward (bias + ELU), 6 = dgrad with B = W^T given K-major
    if (mode == 5) launch_gemm_bf16<EPI_BIAS_ELU>(nullptr, g, st); else launch_gemm_bf16<EPI_ELU_GRAD>(nullptr, g, st);
    HX_CHECK(hipGetLastError());
    return 0;
  }
  if (mode == 8 || mode == 9) {      // forward product on the persistent grid of the rollout's background critic (8: 64-row, 9: 128-row tiles)
    const int grid = 7;              // deliberately small and odd: every workgroup walks many tiles, the last round is ragged
    if (mode == 8) {
      g.tiles_m = (M + 63) / 64; g.tiles_n = (N + 127) / 128;
      hipLaunchKernelGGL((hx_gemm_persistent_kernel<64, 128, HX_BK_ROLL, true, true, EPI_BIAS_ELU>), dim3(grid), dim3(256), 0, st, g, g.tiles_m * g.tiles_n);
    } else {
      g.tiles_m = (M + 127) / 128; g.tiles_n = (N + 127) / 128;
      hipLaunchKernelGGL((hx_gemm_persistent_kernel<128, 128, 16, true, true, EPI_BIAS_ELU>), dim3(grid), dim3(256), 0, st, g, g.tiles_m * g.tiles_n);
    }
    HX_CHECK(hipGetLastError());
    return 0;
  }
  const int variant = mode / 10;     // 0: BK 16, 1: BK 32
  mode %= 10;
#define HX_DISPATCH(BKV)                                                                                   \
  if (mode == 0) launch_gemm<128, 128, BKV, true, true, EPI_BIAS_ELU>(nullptr, g, st);                      \
  else if (mode == 3) launch_gemm<64, 128, BKV, true, true, EPI_BIAS_ELU>(nullptr, g, st);                  \
  else if (mode == 1) launch_gemm<128, 128, BKV, true, false, EPI_ELU_GRAD>(nullptr, g, st);                \
  else if (mode == 4) launch_gemm<64, 128, BKV, true, false, EPI_ELU_GRAD>(nullptr, g, st);                 \
  else if (mode == 2) {                                                                                    \
    /* single split, direct output (C holds [M][ldc]); column sums of A are written to `bias` */           \
    g.splits = 1; g.kchunk = rup(K, 32); g.db_parts = (N + 127) / 128;                                    \
    float* dbtmp = nullptr;                                                                                \
    HX_CHECK(hipMalloc(&dbtmp, (size_t)g.db_parts * M * sizeof(float)));                                   \
    g.dbias = dbtmp;                                                                                       \
    launch_gemm<128, 128, BKV, false, false, EPI_SLAB>(nullptr, g, st);                                     \
    hipLaunchKernelGGL(hx_slab_chunk_kernel, dim3((M + 255) / 256, 1), dim3(256), 0, st, dbtmp, g.db_parts, M, g.db_parts, const_cast<float*>(bias)); \
    HX_CHECK(hipStreamSynchronize(st));                                                                    \
    (void)hipFree(dbtmp);                                                                                  \
  } else { hx_set_error("hx_ppo_gemm_test: bad mode"); return -2; }
  if (variant == 0) { HX_DISPATCH(16) } else { HX_DISPATCH(32) }
#undef HX_DISPATCH
  HX_CHECK(hipGetLastError());
  return 0;
}

// the two words the background critic's sleep hangs on (count of launches it yields to; "a stacking launch raised it"): both are 0
// whenever no rollout launch is in flight -- what tests/test_gpu_runner.py asserts after rollouts of every kind
extern "C" int hx_ppo_pause_words(hx_ppo* s, int32_t* out_h /*[2]*/) {
  if (!s || !out_h) { hx_set_error("hx_ppo_pause_words: null argument"); return -2; }
  out_h[0] = out_h[1] = 0;
  if (!s->pause_flag) return 0;
  HX_CHECK(hipStreamSynchronize(s->stream));
  if (s->stream2) HX_CHECK(hipStreamSynchronize(s->stream2));
  HX_CHECK(hipMemcpy(out_h, s->pause_flag, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return 0;
}
// phase stamps of the fused actor's last launch (-DHX_ACTOR_PROF builds, tools/actor_prof.py): [blocks][8] 100 MHz ticks
extern "C" int hx_ppo_actor_stamps(hx_ppo* s, long long* out_h, int blocks) {
#ifdef HX_ACTOR_PROF
  if (!s || !s->pause_flag || !out_h || blocks < 1 || blocks > 4096) { hx_set_error("hx_ppo_actor_stamps: bad argument"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(out_h, s->pause_flag + 16, (size_t)blocks * 8 * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
#else
  (void)s; (void)out_h; (void)blocks; hx_set_error("hx_ppo_actor_stamps: library built without -DHX_ACTOR_PROF"); return -2;
#endif
}

// the planner's output for `nl` layers, without touching the GPU (tests/test_host_logic.py): rows of
// {layer, col0, ncols, shape, tile rows, tile cols, tiles, splits, kchunk, launch, slab offset (floats), bias slab offset or -1}
extern "C" int hx_wgrad_plan_describe(int nl, const int* out_h, const int* in_ld_h, int rows, int slots, long long* pieces_h, int max_pieces, int* n_pieces,
                                      int* nlaunch_h, long long* slab_floats_h) {
  if (nl < 1 || nl > 6 || !pieces_h || !n_pieces || slots < 1) { hx_set_error("hx_wgrad_plan_describe: bad argument"); return -2; }
  WgradLayerDesc wl[6];
  for (int l = 0; l < nl; ++l) wl[l] = WgradLayerDesc{out_h[l], in_ld_h[l]};
  const WgradPlan plan = hx_wgrad_plan(wl, nl, rows, slots);
  if ((int)plan.pieces.size() > max_pieces) { hx_set_error("hx_wgrad_plan_describe: more pieces than room"); return -2; }
  int k = 0;
  for (const WgradPiece& p : plan.pieces) {
    long long* r = pieces_h + 12 * k++;
    r[0] = p.layer; r[1] = p.col0; r[2] = p.ncols; r[3] = p.shape; r[4] = HX_WSHAPE[p.shape].bm; r[5] = HX_WSHAPE[p.shape].bn; r[6] = p.tiles; r[7] = p.splits;
    r[8] = p.kchunk; r[9] = p.launch; r[10] = (long long)p.slab_off; r[11] = p.bias ? (long long)p.bslab_off : -1;
  }
  *n_pieces = k;
  if (nlaunch_h) *nlaunch_h = plan.nlaunch;
  if (slab_floats_h) *slab_floats_h = (long long)plan.slab_floats;
  return 0;
}

// unit-test hook (tests/test_gpu_gemm.py): the weight gradients of `nl` layers over `rows` rows through the planner, the
// multi-shape kernel and the slab reduction, on caller-supplied device buffers; slots = 0: one workgroup per CU of this device
extern "C" int hx_ppo_wgrad_multi_test(int nl, const int* out_h, const int* in_ld_h, int rows, const float* const* dZ_h, const float* const* X_h,
                                       float* const* dW_h, float* const* db_h, int slots, int* nlaunch_h, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nl < 1 || nl > 6 || rows % 32 != 0) { hx_set_error("hx_ppo_wgrad_multi_test: 1..6 layers, rows a multiple of 32"); return -2; }
  WgradLayerDesc wl[6]; WgradOperands wo[6];
  for (int l = 0; l < nl; ++l) { wl[l] = WgradLayerDesc{out_h[l], in_ld_h[l]}; wo[l] = WgradOperands{dZ_h[l], X_h[l], in_ld_h[l]}; }
  if (slots <= 0) { int dev = 0; slots = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&slots, hipDeviceAttributeMultiprocessorCount, dev); }
  const WgradPlan plan = hx_wgrad_plan(wl, nl, rows, slots);
  if ((int)plan.pieces.size() * 2 > HX_MAX_SEG) { hx_set_error("wgrad plan: too many pieces"); return -2; }
  float *slab = nullptr, *bslab = nullptr;
  HX_CHECK(hipMalloc((void**)&slab, plan.slab_floats * sizeof(float))); HX_CHECK(hipMalloc((void**)&bslab, (plan.bslab_floats + 4) * sizeof(float)));
  HX_CHECK(hipMemsetAsync(slab, 0xff, plan.slab_floats * sizeof(float), st));        // NaN: an element nobody writes shows up in the sums
  HX_CHECK(hipMemsetAsync(bslab, 0xff, plan.bslab_floats * sizeof(float), st));
  for (int q = 0; q < plan.nlaunch; ++q) {
    WgradMulti W;
    hx_wgrad_fill(plan, q, wl, wo, rows, slab, bslab, W);
    hipLaunchKernelGGL(hx_wgrad_multi_kernel, dim3(hx_group_grid(W.G)), dim3(256), 0, st, W);
  }
  ReduceTable rt{}; unsigned blocks = 0;
  for (const WgradPiece& pc : plan.pieces) {
    int k = rt.nseg++;
    rt.src[k] = slab + pc.slab_off; rt.dst[k] = dW_h[pc.layer] + pc.col0; rt.count[k] = (unsigned)wl[pc.layer].out * (unsigned)pc.ncols;
    rt.cols[k] = (unsigned)pc.ncols; rt.ldd[k] = (unsigned)wl[pc.layer].in_ld; rt.S[k] = pc.splits; rt.block0[k] = blocks; blocks += (rt.count[k] + 1023) / 1024;
    if (pc.bias) {
      k = rt.nseg++;
      rt.src[k] = bslab + pc.bslab_off; rt.dst[k] = db_h[pc.layer]; rt.count[k] = (unsigned)wl[pc.layer].out; rt.cols[k] = rt.count[k]; rt.ldd[k] = rt.count[k];
      rt.S[k] = pc.splits * pc.tiles_n; rt.block0[k] = blocks; blocks += (wl[pc.layer].out + 1023) / 1024;
    }
  }
  rt.block0[rt.nseg] = blocks;
  hipLaunchKernelGGL(hx_reduce_all_kernel, dim3(blocks), dim3(256), 0, st, rt);
  HX_CHECK(hipGetLastError());
  HX_CHECK(hipStreamSynchronize(st));
  (void)hipFree(slab); (void)hipFree(bslab);
  if (nlaunch_h) *nlaunch_h = plan.nlaunch;
  return 0;
}

// Matrix-pipe ceiling probe (tools/mfma_peak.py): what rate does v_mfma_f32_32x32x2_f32 sustain when a wave does
//   mode 0  nothing else (register operands, NACC independent accumulators)
//   mode 1  + the GEMM's LDS fragment reads: 4 x ds_read_b128 per 16 MFMAs (2x2 tiles of 32x32, 8-deep k block)
//   mode 2  + one workgroup barrier per 32 MFMAs (a BK = 16 tile)
//   mode 3  + the tile's global loads (4 x dwordx4 per thread) and LDS stores per 32 MFMAs, double-buffered like hx_gemm.h
//   mode 4  = 3 with the wgrad kernel's fragment reads (operands stored [k][rows]: 4 x ds_read_b32 per operand per k block)
//   mode 5  = 3 with guarded global loads (exec-masked, as at the edges of a real matrix; the guard is always true here)
// No result of the probe is meaningful; only the instruction streams are.
template <int MODE>
__global__ void __launch_bounds__(256) hx_mfma_probe_kernel(float* out, const float* __restrict__ src, int n) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * 20];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r32 = lane & 31;
  for (int i = tid; i < 2 * 2 * 128 * 20; i += 256) lds[i] = 1.0f + 1e-6f * (float)(i & 1023);
  __syncthreads();
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  f32x4 fa[2], fb[2];
  fa[0] = fa[1] = fb[0] = fb[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
  f32x4 st[4];
  const float* gp = src + (size_t)blockIdx.x * 4 * 4096 + tid * 4;
  if (MODE >= 3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) st[q] = *reinterpret_cast<const f32x4*>(gp + q * 1024);
  }
  const int wm = wave >> 1, wn = wave & 1;
  for (int it = 0; it < n; it += 32) {                  // one BK = 16 tile: 2 k-blocks x 16 MFMAs
    const int buf = (it >> 5) & 1;
    const float* As = lds + buf * (2 * 128 * 20);
    const float* Bs = As + 128 * 20;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            fa[i][j] = As[(kb * 8 + 4 * h + j) * 128 + wm * 64 + i * 32 + r32];
            fb[i][j] = Bs[(kb * 8 + 4 * h + j) * 128 + wn * 64 + i * 32 + r32];
          }
      } else if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const f32x4*>(As + (wm * 64 + i * 32 + r32) * 20 + kb * 8 + 4 * h);
          fb[i] = *reinterpret_cast<const f32x4*>(Bs + (wn * 64 + i * 32 + r32) * 20 + kb * 8 + 4 * h);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
    if (MODE >= 3) {
      float* Ws = lds + (buf ^ 1) * (2 * 128 * 20);
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(Ws + ((tid + q * 256) >> 2) * 20 + ((tid + q * 256) & 3) * 4) = st[q];
      const float* gq = gp + (size_t)(((it >> 5) + 1) & 3) * 4096;     // walks a 64 KB window per workgroup: L2 / Infinity-Cache hits, and the loads cannot be hoisted out of the loop
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MODE == 5) { st[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (tid * 4 + q < n) st[q] = *reinterpret_cast<const f32x4*>(gq + q * 1024); }
        else st[q] = *reinterpret_cast<const f32x4*>(gq + q * 1024);
      }
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  out[blockIdx.x * 256 + tid] = s;
}
extern "C" int hx_mfma_probe(int mode, int blocks, int n, float* tflops_out) {
  float *out = nullptr, *src = nullptr;
  HX_CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  HX_CHECK(hipMalloc(&src, (size_t)blocks * 4 * 4096 * 4));
  HX_CHECK(hipMemset(src, 0x3c, (size_t)blocks * 4 * 4096 * 4));
  hipEvent_t e0, e1; HX_CHECK(hipEventCreate(&e0)); HX_CHECK(hipEventCreate(&e1));
  auto run = [&]() {
    if (mode == 0) hipLaunchKernelGGL(hx_mfma_probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, src, n);
    else if (mode == 1) hipLaunchKernelGGL(hx_mfma_probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, src, n);
    else if (mode == 2) hipLaunchKernelGGL(hx_mfma_probe_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, src, n);
    else if (mode == 3) hipLaunchKernelGGL(hx_mfma_probe_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, src, n);
    else if (mode == 4) hipLaunchKernelGGL(hx_mfma_probe_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, src, n);
    else hipLaunchKernelGGL(hx_mfma_probe_kernel<5>, dim3(blocks), dim3(256), 0, 0, out, src, n);
  };
  run();
  HX_CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) run();
  HX_CHECK(hipEventRecord(e1, 0));
  HX_CHECK(hipEventSynchronize(e1));
  HX_CHECK(hipGetLastError());
  float ms = 0; HX_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *tflops_out = (float)(5.0 * blocks * 4.0 * n * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12);
  (void)hipFree(out); (void)hipFree(src); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

// timing hook (tools/gemm_bench.py): `iters` launches of one learner GEMM on scratch buffers filled with a
// non-trivial bit pattern; returns the mean milliseconds per launch measured with HIP events on the stream.
extern "C" int hx_ppo_gemm_bench(int kind, int bk, int rows, int out, int in_ld, int iters, float* ms_out) {
  // kind 0: fwd  Y[rows][out] = X[rows][in_ld] W[out][in_ld]^T ; 1: dgrad dX[rows][in_ld] = dZ[rows][out] W[out][in_ld]
  // kind 2: wgrad dW[out][in_ld] = dZ[rows][out]^T X[rows][in_ld] with the production split-K
  float *X, *W, *Y, *slab, *bslab;
  const size_t nx = (size_t)rows * in_ld, ny = (size_t)rows * out, nw = (size_t)out * in_ld;
  HX_CHECK(hipMalloc(&X, nx * 4)); HX_CHECK(hipMalloc(&W, nw * 4)); HX_CHECK(hipMalloc(&Y, ny * 4));
  const int tiles = ((out + 127) / 128) * ((in_ld + 127) / 128);
  static int target_blocks = -1, round_up = 1;
  if (target_blocks < 0) {
    const char* e = getenv("HX_WGRAD_BLOCKS");
    target_blocks = e ? atoi(e) : 1024;
    round_up = getenv("HX_WGRAD_FLOOR") ? 0 : 1;
  }
  int splits = round_up ? (target_blocks + tiles - 1) / tiles : target_blocks / tiles;
  if (splits < 1) splits = 1; int max_splits = rows / HX_WGRAD_MIN_CHUNK; if (max_splits < 1) max_splits = 1; if (splits > max_splits) splits = max_splits;
  int kchunk = rup((rows + splits - 1) / splits, 32); splits = (rows + kchunk - 1) / kchunk;
  HX_CHECK(hipMalloc(&slab, (size_t)splits * nw * 4)); HX_CHECK(hipMalloc(&bslab, (size_t)splits * ((in_ld + 127) / 128) * out * 4));
  HX_CHECK(hipMemset(X, 0x3d, nx * 4)); HX_CHECK(hipMemset(W, 0x3c, nw * 4)); HX_CHECK(hipMemset(Y, 0x3b, ny * 4));
  hipStream_t st; HX_CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; HX_CHECK(hipEventCreate(&e0)); HX_CHECK(hipEventCreate(&e1));
  const bool bf16_bench = (bk == 200);
  const int bm = bk / 100 ? 64 : 128; bk %= 100;
  auto run = [&]() {
    GemmArgs g{};
#define HX_V(BMV, BKV, AK, BKM, EPI) launch_gemm<BMV, 128, BKV, AK, BKM, EPI>(nullptr, g, st)
#define HX_PICK(AK, BKM, EPI) do { if (bm == 128) { if (bk == 16) HX_V(128, 16, AK, BKM, EPI); else HX_V(128, 32, AK, BKM, EPI); } \
                                   else { if (bk == 16) HX_V(64, 16, AK, BKM, EPI); else HX_V(64, 32, AK, BKM, EPI); } } while (0)
    if (bf16_bench && kind == 0) { g.A = X; g.lda = in_ld; g.B = W; g.ldb = in_ld; g.C = Y; g.ldc = out; g.M = rows; g.N = out; g.K = in_ld; g.bias = bslab;
      launch_gemm_bf16<EPI_BIAS_ELU>(nullptr, g, st); }
    else if (bf16_bench && kind == 1) { g.A = Y; g.lda = out; g.B = W; g.ldb = out; g.C = X; g.ldc = in_ld; g.M = rows; g.N = in_ld; g.K = out; g.H = X; g.ldh = in_ld;
      launch_gemm_bf16<EPI_ELU_GRAD>(nullptr, g, st); }      // W buffer read as W^T[in_ld][out]: same byte count
    else if (kind == 0) { g.A = X; g.lda = in_ld; g.B = W; g.ldb = in_ld; g.C = Y; g.ldc = out; g.M = rows; g.N = out; g.K = in_ld; g.bias = bslab;
      HX_PICK(true, true, EPI_BIAS_ELU); }
    else if (kind == 1) { g.A = Y; g.lda = out; g.B = W; g.ldb = in_ld; g.C = X; g.ldc = in_ld; g.M = rows; g.N = in_ld; g.K = out; g.H = X; g.ldh = in_ld;
      HX_PICK(true, false, EPI_ELU_GRAD); }
    else { g.A = Y; g.lda = out; g.B = X; g.ldb = in_ld; g.C = slab; g.ldc = in_ld; g.M = out; g.N = in_ld; g.K = rows; g.splits = splits; g.kchunk = kchunk; g.dbias = bslab; g.db_parts = (in_ld + 127) / 128;
      if (getenv("HX_BENCH_LD0")) { g.lda = 0; g.ldb = 0; }      // experiment: every k row aliases row 0 -> operands come from the L1
      if (getenv("HX_BENCH_NODB")) g.dbias = nullptr;             // experiment: without the bias-gradient column sums
      if (bk == 16) { if (rows % 16 == 0 && !getenv("HX_BENCH_NOKFULL")) launch_gemm<128, 128, 16, false, false, EPI_SLAB, true>(nullptr, g, st); else HX_V(128, 16, false, false, EPI_SLAB); }
      else HX_V(128, 32, false, false, EPI_SLAB); }
#undef HX_PICK
#undef HX_V
  };
  for (int i = 0; i < 3; ++i) run();
  HX_CHECK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) run();
  HX_CHECK(hipEventRecord(e1, st));
  HX_CHECK(hipStreamSynchronize(st));
  float ms = 0; HX_CHECK(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(slab); (void)hipFree(bslab);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
  return 0;
}

static int ppo_create_impl(const hx_ppo_cfg* cfg, void* stream, void* ext_grad, hx_ppo* s);
extern "C" void hx_ppo_destroy(hx_ppo* s);
extern "C" int hx_ppo_create(const hx_ppo_cfg* cfg, void* stream, void* ext_grad, hx_ppo** out) {
  if (int rc = hx_knobs_check()) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { hx_set_error("hx_ppo_create: no HIP device (this library has no CPU path)"); return -1; }
  if (cfg->num_actions > MAX_A || cfg->num_actions < 1) { hx_set_error("hx_ppo_create: num_actions must be in [1, 32]"); return -2; }
  for (int l = 0; l < 3; ++l)
    if (cfg->actor_hidden[l] % 4 != 0 || cfg->critic_hidden[l] % 4 != 0 || cfg->actor_hidden[l] <= 0 || cfg->critic_hidden[l] <= 0) {
      hx_set_error("hx_ppo_create: hidden layer widths must be positive multiples of 4 (16-byte vector accesses)"); return -2;
    }
  if (cfg->actor_hidden[2] % 64 || cfg->critic_hidden[2] % 64) { hx_set_error("hx_ppo_create: last hidden widths must be multiples of 64 (head kernels split a row over 16 lanes of 4-wide loads)"); return -2; }
  {
    const int A_ = cfg->num_actions, ha_ = cfg->actor_hidden[2], hc_ = cfg->critic_hidden[2];
    const size_t head_lds = (size_t)(HEAD_ROWS * (ha_ + 1) + HEAD_ROWS * (hc_ + 1) + A_ * ha_ + hc_ + HEAD_ROWS * (A_ + 1) + HEAD_ROWS * (4 + A_)) * sizeof(float);
    if (head_lds > 160 * 1024 || (size_t)(A_ * ha_ + hc_) * sizeof(float) > 64 * 1024) { hx_set_error("hx_ppo_create: last hidden widths too large for the loss head's LDS tile"); return -2; }
  }
  static_assert(HEAD_ROWS * 8 == 256, "loss head: 8 lanes per row");
  hx_ppo* s = new hx_ppo();
  const int rc = ppo_create_impl(cfg, stream, ext_grad, s);
  if (rc) { hx_ppo_destroy(s); return rc; }      // nothing of a half-built learner survives an error
  *out = s;
  return 0;
}

static int ppo_create_impl(const hx_ppo_cfg* cfg, void* stream, void* ext_grad, hx_ppo* s) {
  s->cfg = *cfg;
  if (int rc = hx_knobs_check()) return rc;
  int knob_streams = 1, knob_bg_persist = -1, knob_wgrad = 0;
  bool cu_set = false; unsigned cu_word = 0;
  if (int rc = hx_knob_hex32("HX_CRITIC_CU_WORD", &cu_set, &cu_word)) return rc;
  if (int rc = hx_knob_int("HX_UPDATE_STREAMS", 1, 1, 2, &knob_streams)) return rc;
  if (int rc = hx_knob_int("HX_ACTOR_WAVES", 8, 2, 8, &s->actor_waves)) return rc;
  if (s->actor_waves != 2 && s->actor_waves != 4 && s->actor_waves != 8) { hx_set_error("HX_ACTOR_WAVES: 2 (probe, fp32 rows only), 4 or 8"); return -2; }
  if (int rc = hx_knob_int("HX_ACTOR_DEPTH", 2, 2, 6, &s->actor_depth)) return rc;
  if (s->actor_depth == 5) { hx_set_error("HX_ACTOR_DEPTH: 2, 3, 4 or 6"); return -2; }
  if (int rc = hx_knob_int("HX_ACTOR_ROWS", 0, 0, 32, &s->actor_rows)) return rc;
  if (s->actor_rows != 0 && s->actor_rows != 16 && s->actor_rows != 32) { hx_set_error("HX_ACTOR_ROWS: 0 (by batch size), 16 or 32"); return -2; }
  if (int rc = hx_knob_int("HX_CRITIC_LATE", 0, 0, 1, &s->critic_late)) return rc;
  if (int rc = hx_knob_int("HX_BG_PERSIST", -1, 0, 4096, &knob_bg_persist)) return rc;
  if (int rc = hx_knob_int("HX_FWD_IN_TILE", 128, 64, 128, &s->fwd_in_tile)) return rc;
  if (s->fwd_in_tile != 64 && s->fwd_in_tile != 128) { hx_set_error("HX_FWD_IN_TILE: 64 or 128"); return -2; }
  if (int rc = hx_knob_int("HX_BG_TILE", 0, 0, 128, &s->bg_tile)) return rc;
  if (int rc = hx_knob_int("HX_STACK_PAUSE", 1, 0, 1, &s->stack_pause)) return rc;
  if (int rc = hx_knob_int("HX_BG_WAVES", 4, 2, 4, &s->bg_waves)) return rc;
  if (s->bg_waves == 3) { hx_set_error("HX_BG_WAVES: 2 or 4"); return -2; }
  if (s->bg_tile != 0 && s->bg_tile != 64 && s->bg_tile != 128) { hx_set_error("HX_BG_TILE: 0 (by batch size), 64 or 128"); return -2; }
  if (int rc = hx_knob_int("HX_CRITIC_CHUNK", HX_CRITIC_CHUNK, 1, 4096, &s->critic_chunk)) return rc;
  if (int rc = hx_knob_int("HX_WGRAD_BLOCKS", 0, 1, 65536, &knob_wgrad)) return rc;
  if (int rc = hx_knob_int("HX_WGRAD_GROUP", 1, 0, 1, &s->wgrad_group)) return rc;
  if (int rc = hx_knob_int("HX_WGRAD_MULTI", 1, 0, 1, &s->wgrad_multi)) return rc;
  if (int rc = hx_knob_int("HX_GEMM_PAIR", 1, 0, 1, &s->gemm_pair)) return rc;
  if (int rc = hx_knob_int("HX_GEMM_SP", 1, 0, 1, &s->gemm_sp)) return rc;
  if (int rc = hx_knob_int("HX_FRAMES_GATHER", -1, 0, 1, &s->frames_gather)) return rc;
  // by measurement (profiles/r04_o_frames.txt): expanded minibatch rows up to 8192 robots, gathered loaders above (no expanded copy: 6.6 GB at 16 384)
  if (s->frames_gather < 0) s->frames_gather = (cfg->num_envs >= 16384) ? 1 : 0;
  if (int rc = hx_knob_int("HX_HEAD_MFMA", 1, 0, 1, &s->head_mfma)) return rc;
  int knob_yield = 1;
  if (int rc = hx_knob_int("HX_CRITIC_YIELD", 1, 0, 1, &knob_yield)) return rc;
  if (stream) { s->stream = (hipStream_t)stream; s->own_stream = false; }
  else { HX_CHECK(hipStreamCreate(&s->stream)); s->own_stream = true; }
  {
    // The deferred critic is background work: lowest stream priority, so the rollout's critical path (actor kernel,
    // env-step kernel) is dispatched first whenever both have workgroups waiting.  Confining it to a subset of the
    // CUs (hipExtStreamCreateWithCUMask) was measured and is WORSE (profiles/r01_critic_cu_mask.txt): the env-step
    // kernel's slow-down during critic bursts is not a CU-occupancy effect, and a narrower critic only finishes later.
    // HX_CRITIC_CU_WORD=<hex 32-bit word, repeated for each group of 32 CUs> keeps the experiment reproducible.
    if (cu_set) {
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = cu_word;
      HX_CHECK(hipExtStreamCreateWithCUMask(&s->stream2, 8, mask));
    } else {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      HX_CHECK(hipStreamCreateWithPriority(&s->stream2, hipStreamNonBlocking, least));
    }
  }
  // measured (profiles/r02_c_update_phase.txt): no gain at 4096 envs (31.70 ms against 31.41 ms on one stream) -- the launches
  // of one chain already keep every CU busy; kept behind HX_UPDATE_STREAMS=2 for other sizes
  if (knob_streams == 2) {
    HX_CHECK(hipStreamCreateWithFlags(&s->stream_b, hipStreamNonBlocking));
    HX_CHECK(hipEventCreateWithFlags(&s->ev_b0, hipEventDisableTiming));
    HX_CHECK(hipEventCreateWithFlags(&s->ev_b1, hipEventDisableTiming));
  }
  HX_CHECK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  HX_CHECK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  HX_CHECK(hipEventCreateWithFlags(&s->ev_priv, hipEventDisableTiming));
  HX_CHECK(hipEventCreateWithFlags(&s->ev_crit, hipEventDisableTiming));
  HX_CHECK(hipEventRecord(s->ev_crit, s->stream2));
  s->crit_done = 0;
  const int A = cfg->num_actions, N = cfg->num_envs, T = cfg->num_steps;
  // ---- parameter layout
  int dims_a[5] = {cfg->num_obs, cfg->actor_hidden[0], cfg->actor_hidden[1], cfg->actor_hidden[2], A};
  int dims_c[5] = {cfg->num_priv, cfg->critic_hidden[0], cfg->critic_hidden[1], cfg->critic_hidden[2], 1};
  size_t off = 0; int64_t tc = A;
  for (int net = 0; net < 2; ++net)
    for (int l = 0; l < 4; ++l) {
      const int* d = net ? dims_c : dims_a;
      Layer& Ly = s->L[net * 4 + l];
      Ly.in = d[l]; Ly.out = d[l + 1]; Ly.in_ld = rup(d[l], 4);
      Ly.w = off; off += (size_t)Ly.out * Ly.in_ld; off = rup((int)off, 4);
      Ly.b = off; off += Ly.out; off = rup((int)off, 4);
      tc += (int64_t)Ly.out * Ly.in + Ly.out;
    }
  s->std_off = off; off += rup(A, 4);
  s->padded = off; s->stats_off = off;
  s->torch_count = tc;
  if (cfg->obs_ld < s->L[0].in_ld || cfg->priv_ld < s->L[4].in_ld || cfg->obs_ld % 4 || cfg->priv_ld % 4) { hx_set_error("hx_ppo_create: obs_ld/priv_ld must be >= padded input width and multiples of 4"); return -2; }
  int rc = 0;
  rc |= palloc(s, &s->params, s->padded);
  if (ext_grad) { s->grads = (float*)ext_grad; s->ext_grads = true; }
  else { rc |= palloc(s, &s->grads, s->padded + 4); s->ext_grads = false; }
  rc |= palloc(s, &s->m, s->padded);
  rc |= palloc(s, &s->v, s->padded);
  // ---- storage
  const size_t TN = (size_t)T * N;
  s->frames = cfg->obs_frame > 0 || cfg->priv_frame > 0 || cfg->obs_stack > 0 || cfg->priv_stack > 0;
  if (s->frames) {
    if (cfg->obs_frame * cfg->obs_stack != cfg->num_obs || cfg->priv_frame * cfg->priv_stack != cfg->num_priv || cfg->obs_frame < 1 || cfg->priv_frame < 1) {
      hx_set_error("hx_ppo_create: frame storage needs obs_frame * obs_stack == num_obs and priv_frame * priv_stack == num_priv"); return -2;
    }
    // the per-row "frames since reset" count saturates at obs_stack and the privileged rows' zero prefix is derived from the same
    // count (hx_env.h env_glue, hx_stack_io_kernel): a longer privileged stack would keep its oldest frames masked for ever
    if (cfg->priv_stack > cfg->obs_stack) { hx_set_error("hx_ppo_create: frame storage needs priv_stack <= obs_stack (the reset age is counted up to obs_stack)"); return -2; }
    s->fo = cfg->obs_frame; s->fp = cfg->priv_frame; s->So = cfg->obs_stack; s->Sp = cfg->priv_stack; s->Po = T + s->So; s->Pp = T + s->Sp;
    const size_t no = (size_t)N * s->Po * s->fo, np = (size_t)N * s->Pp * s->fp;
    if (no + 64 >= (1ull << 31) || np + 64 >= (1ull << 31)) { hx_set_error("hx_ppo_create: frame rings beyond 2^31 floats (row starts are 32-bit offsets)"); return -2; }
    // + slack: the 16-byte loads of a row's last tile and the padding columns reach a few floats past the last ring; those
    // elements are masked but must be readable and finite (the buffers are zero-filled)
    rc |= palloc(s, &s->s_obs, no + 64);
    rc |= palloc(s, &s->s_priv, np + 64);
    rc |= palloc(s, &s->off_obs, (TN + N)); rc |= palloc(s, &s->off_priv, (TN + N));
    rc |= palloc(s, &s->kz_obs, (TN + N)); rc |= palloc(s, &s->kz_priv, (TN + N));
    rc |= palloc(s, &s->mb_off_obs, TN); rc |= palloc(s, &s->mb_off_priv, TN); rc |= palloc(s, &s->mb_kz_obs, TN); rc |= palloc(s, &s->mb_kz_priv, TN);
    if (rc) return -3;
    std::vector<int> oo(TN + N), op(TN + N);
    for (int t = 0; t <= T; ++t)
      for (int e = 0; e < N; ++e) { oo[(size_t)t * N + e] = (int)(((size_t)e * s->Po + t) * s->fo); op[(size_t)t * N + e] = (int)(((size_t)e * s->Pp + t) * s->fp); }
    HX_CHECK(hipMemcpyAsync(s->off_obs, oo.data(), oo.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
    HX_CHECK(hipMemcpyAsync(s->off_priv, op.data(), op.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
    HX_CHECK(hipStreamSynchronize(s->stream));
  } else {
    rc |= palloc(s, &s->s_obs, TN * cfg->obs_ld);
    rc |= palloc(s, &s->s_priv, TN * cfg->priv_ld);
  }
  rc |= palloc(s, &s->s_actions, TN * A); rc |= palloc(s, &s->s_mu, TN * A);
  rc |= palloc(s, &s->s_values, TN); rc |= palloc(s, &s->s_logp, TN); rc |= palloc(s, &s->s_rewards, TN);
  rc |= palloc(s, &s->s_returns, TN); rc |= palloc(s, &s->s_adv_raw, TN); rc |= palloc(s, &s->s_adv, TN);
  rc |= palloc(s, &s->s_dones, TN); rc |= palloc(s, &s->s_timeouts, TN); rc |= palloc(s, &s->sigma_old, MAX_A);
  rc |= palloc(s, &s->last_values, (size_t)N); rc |= palloc(s, &s->moments, 3);
  // ---- workspace
  const int mbs = (int)(TN / cfg->num_mini_batches);
  s->Mmax = mbs > N ? mbs : N;
  const size_t Mm = s->Mmax;
  // The reference reuses ONE permutation for every epoch (rollout_storage.py:149 draws it outside the epoch loop), so
  // minibatch i holds the same rows in each epoch: with more than one epoch the gathered rows are kept per minibatch
  // (a permuted copy of the rollout, 1.6 GB at 4096 envs) and gathered once per update instead of once per epoch.
  s->mb_slots = (cfg->num_learning_epochs > 1 && cfg->num_mini_batches <= 64 && mbs >= N) ? cfg->num_mini_batches : 1;
  if (s->frames && !s->frames_gather) {
    // rows of a minibatch expanded once per update and kept across the epochs (hx_expand_mb_kernel): the permuted copy row storage
    // keeps, without the unpermuted one; everything after the expansion is the plain path
    rc |= palloc(s, &s->obs_mb_all, (size_t)s->mb_slots * Mm * cfg->obs_ld); rc |= palloc(s, &s->priv_mb_all, (size_t)s->mb_slots * Mm * cfg->priv_ld);
    rc |= palloc(s, &s->row_mb_all, TN * (2 * A + 4));
  } else if (s->frames) {        // no permuted copy of the rollout: tables in permutation order (hx_mb_tables_kernel) + one minibatch of rows
    s->mb_slots = 1;
    rc |= palloc(s, &s->obs_mb_all, Mm * cfg->obs_ld); rc |= palloc(s, &s->priv_mb_all, Mm * cfg->priv_ld);
    rc |= palloc(s, &s->row_mb_all, TN * (2 * A + 4));
  } else {
    rc |= palloc(s, &s->obs_mb_all, (size_t)s->mb_slots * Mm * cfg->obs_ld); rc |= palloc(s, &s->priv_mb_all, (size_t)s->mb_slots * Mm * cfg->priv_ld);
    rc |= palloc(s, &s->row_mb_all, (size_t)s->mb_slots * Mm * (2 * A + 4));
  }
  s->obs_mb = s->obs_mb_all; s->priv_mb = s->priv_mb_all; s->row_mb = s->row_mb_all;
  for (int l = 0; l < 3; ++l) {
    rc |= palloc(s, &s->act_a[l], Mm * cfg->actor_hidden[l]); rc |= palloc(s, &s->dz_a[l], Mm * cfg->actor_hidden[l]);
    rc |= palloc(s, &s->act_c[l], Mm * cfg->critic_hidden[l]); rc |= palloc(s, &s->dz_c[l], Mm * cfg->critic_hidden[l]);
  }
  size_t slab_tot = 0, bslab_tot = 0;
  for (int i = 0; i < 8; ++i) {
    s->slab_off[i] = slab_tot; s->bslab_off[i] = bslab_tot;
    if (i % 4 == 3) continue;
    const Layer& Ly = s->L[i];
    const int tiles = ((Ly.out + 127) / 128) * ((Ly.in_ld + 127) / 128);
    int splits = (1024 + tiles - 1) / tiles + 1;
    int max_splits = s->Mmax / 256; if (max_splits < 1) max_splits = 1;
    if (splits > max_splits + 1) splits = max_splits + 1;
    s->slab_splits[i] = splits;
    slab_tot += (size_t)splits * Ly.out * Ly.in_ld;
    bslab_tot += (size_t)splits * ((Ly.in_ld + 127) / 128) * Ly.out;
  }
  s->slab_floats = slab_tot;
  rc |= palloc(s, &s->slab, slab_tot);
  rc |= palloc(s, &s->bias_slab, bslab_tot);
  s->head_slab_w = A * cfg->actor_hidden[2] + A + cfg->critic_hidden[2] + 1 + A + 4;
  s->head_blocks_max = (s->Mmax + HEAD_ROWS - 1) / HEAD_ROWS;
  rc |= palloc(s, &s->head_slab, (size_t)s->head_blocks_max * s->head_slab_w);
  rc |= palloc(s, &s->head_slab2, (size_t)((s->head_blocks_max + 31) / 32) * s->head_slab_w);
  rc |= palloc(s, &s->perm, TN);
#ifdef HX_ACTOR_PROF
  if (knob_yield) rc |= palloc(s, &s->pause_flag, 16 + 4096 * 16);      // + phase stamps of the fused actor's workgroups
#else
  if (knob_yield) rc |= palloc(s, &s->pause_flag, 16);      // zeroed; only element 0 is used
#endif
  rc |= palloc(s, &s->sumsq, 1); rc |= palloc(s, &s->sched, 1);
  if (rc) return -3;
  SchedState st0{cfg->learning_rate, 0.f, 0.f, 0.f};
  HX_CHECK(hipMemcpyAsync(s->sched, &st0, sizeof(st0), hipMemcpyHostToDevice, s->stream));
  // std = init_noise_std (actor_critic.py:80); weights stay zero until hx_ppo_set_params_h
  std::vector<float> sd(rup(A, 4), 0.f);
  for (int j = 0; j < A; ++j) sd[j] = cfg->init_noise_std;
  HX_CHECK(hipMemcpyAsync(s->params + s->std_off, sd.data(), sd.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 8, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 8, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<false, 8, 1, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  HX_CHECK(hipFuncSetAttribute((const void*)hx_actor_fused_kernel<true, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  // 8 waves (each streams 1/8 of a layer's weight rows) keep twice the bytes in flight per CU: 68 us per call against
  // 74 us with 4 waves at 4096 rows (profiles/r01_g_actor_ring.txt); results are bitwise the same.  HX_ACTOR_WAVES=4 for A/B runs.
  for (int l = 0; l < 3; ++l) s->apack[l] = nullptr;
  s->apack_dirty = true;
  {
    // Background critic on a persistent grid of HALF the CUs (one workgroup each): the env-step kernel's waves need a
    // whole SIMD's registers, so they can only land on CUs that hold no GEMM wave at all; 4096 robots = 512 such waves =
    // the SIMDs of the other half.  More robots than that need every CU for the env step itself: ordinary launches then.
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    s->bg_persist = knob_bg_persist >= 0 ? knob_bg_persist : ((cfg->num_envs <= 16 * cus) ? (s->bg_waves == 2 ? cus : cus / 2) : 0);
  }
  {
    const int ha_ = cfg->actor_hidden[2], hc_ = cfg->critic_hidden[2];
    const size_t head_lds = head_lds_bytes(ha_, hc_, A, head_lds_bytes(ha_, hc_, A, true) <= 64 * 1024);
    if (head_lds > 64 * 1024) {      // a wide ACTOR last layer (the critic is then not staged); hector's 43 KB launch keeps the default
      HX_CHECK(hipFuncSetAttribute((const void*)hx_loss_head_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HX_CHECK(hipFuncSetAttribute((const void*)hx_loss_head_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HX_CHECK(hipFuncSetAttribute((const void*)hx_loss_head_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
  }
  s->step = 0; s->adam_t = 0; s->mb_done = 0; s->mb_total = 0;
  s->seed_lo = 0x1234567u; s->seed_hi = 0x89abcdefu; s->act_counter = 0; s->perm_counter = 0; s->perm_key = 0xA511E9B3u;
  s->prof = false; s->ev_used = 0; s->prof_only = -1;
  return 0;
}

extern "C" void hx_ppo_destroy(hx_ppo* s) {
  if (!s) return;
  (void)hipDeviceSynchronize();      // the stream may be borrowed from an env that no longer exists
  for (void* a : s->allocs) (void)hipFree(a);
  if (s->wslab) (void)hipFree(s->wslab);
  if (s->wbslab) (void)hipFree(s->wbslab);
  for (auto e : s->ev) (void)hipEventDestroy(e);
  // null checks: hx_ppo_create also ends here with a half-built object, and a failed destroy call would leave a
  // sticky HIP error for the next launch check to trip over
  if (s->stream2) { (void)hipStreamSynchronize(s->stream2); (void)hipStreamDestroy(s->stream2); }
  if (s->stream_b) { (void)hipStreamSynchronize(s->stream_b); (void)hipStreamDestroy(s->stream_b); }
  for (hipEvent_t e : {s->ev_b0, s->ev_b1}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {s->ev_fork, s->ev_join, s->ev_priv, s->ev_crit})
    if (e) (void)hipEventDestroy(e);
  if (s->own_stream && s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

extern "C" int hx_ppo_set_comm(hx_ppo* s, hx_comm* c) {
  if (!s) { hx_set_error("hx_ppo_set_comm: null learner"); return -2; }
  s->comm = c;
  return 0;
}
extern "C" int hx_ppo_broadcast_params(hx_ppo* s, int root) {
  if (!s || !s->comm) { hx_set_error("hx_ppo_broadcast_params: no communicator set"); return -2; }
  int rc = hx_comm_broadcast(s->comm, s->params, s->padded, root, s->stream); if (rc) return rc;
  s->apack_dirty = true;
  refresh_transposes(s, s->stream);
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int64_t hx_ppo_num_params(hx_ppo* s) { return s->torch_count; }
// Keys of the learner's two random streams.  sample_seed keys the action noise of PPO.act (Philox counter = env row,
// act call, action index): data-parallel ranks pass seed + rank so that their exploration noise is independent.
// perm_seed keys the minibatch permutation (may be equal on all ranks).  Without this call the keys are fixed constants.
static uint32_t mix32(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return (uint32_t)x; }
extern "C" int hx_ppo_set_seed(hx_ppo* s, uint64_t sample_seed, uint64_t perm_seed) {
  if (!s) { hx_set_error("hx_ppo_set_seed: null learner"); return -2; }
  s->seed_lo = 0x1234567u ^ mix32(sample_seed); s->seed_hi = 0x89abcdefu ^ mix32(sample_seed + 0x9E3779B97F4A7C15ULL);
  s->perm_key = 0xA511E9B3u ^ mix32(perm_seed ^ 0xD1B54A32D192ED03ULL);
  return 0;
}
// Ranks that split ONE logical batch (rank r owns rows [base_r, base_r + num_envs)) pass their first global row: the action
// noise is then a function of the GLOBAL row, and with equal sample seeds the job samples exactly what a single process on
// the union of the shards would.  Default 0 (weak scaling: every rank its own robots, keyed by seed + rank instead).
extern "C" int hx_ppo_set_row_base(hx_ppo* s, uint32_t base) {
  if (!s) { hx_set_error("hx_ppo_set_row_base: null learner"); return -2; }
  s->row_base = base; return 0;
}
// positions in the two streams (checkpointed so that a resumed run does not replay the noise from counter 0)
extern "C" int hx_ppo_get_rng_state(hx_ppo* s, uint32_t* act_counter, uint32_t* perm_counter) {
  if (!s) { hx_set_error("hx_ppo_get_rng_state: null learner"); return -2; }
  *act_counter = s->act_counter; *perm_counter = s->perm_counter; return 0;
}
extern "C" int hx_ppo_set_rng_state(hx_ppo* s, uint32_t act_counter, uint32_t perm_counter) {
  if (!s) { hx_set_error("hx_ppo_set_rng_state: null learner"); return -2; }
  s->act_counter = act_counter; s->perm_counter = perm_counter; return 0;
}
extern "C" void* hx_ppo_stream(hx_ppo* s) { return (void*)s->stream; }

// torch parameters() order <-> padded device layout
template <typename F> static void for_each_tensor(hx_ppo* s, F f) {
  size_t t = 0;
  f(t, s->std_off, 1, s->cfg.num_actions, s->cfg.num_actions); t += s->cfg.num_actions;
  for (int i = 0; i < 8; ++i) {
    const Layer& Ly = s->L[i];
    f(t, Ly.w, Ly.out, Ly.in, Ly.in_ld); t += (size_t)Ly.out * Ly.in;
    f(t, Ly.b, 1, Ly.out, Ly.out); t += Ly.out;
  }
}
static void pack_padded(hx_ppo* s, const float* flat, std::vector<float>& pad) {
  pad.assign(s->padded, 0.f);
  for_each_tensor(s, [&](size_t t, size_t off, int rows, int cols, int ld) {
    for (int r = 0; r < rows; ++r) memcpy(&pad[off + (size_t)r * ld], flat + t + (size_t)r * cols, cols * sizeof(float));
  });
}
static void unpack_padded(hx_ppo* s, const std::vector<float>& pad, float* flat) {
  for_each_tensor(s, [&](size_t t, size_t off, int rows, int cols, int ld) {
    for (int r = 0; r < rows; ++r) memcpy(flat + t + (size_t)r * cols, &pad[off + (size_t)r * ld], cols * sizeof(float));
  });
}
static int upload_flat(hx_ppo* s, float* dst, const float* flat) {
  std::vector<float> pad; pack_padded(s, flat, pad);
  if (int rc = learner_sync(s)) return rc;
  HX_CHECK(hipMemcpy(dst, pad.data(), s->padded * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}
static int download_flat(hx_ppo* s, const float* src, float* flat) {
  std::vector<float> pad(s->padded);
  if (int rc = learner_sync(s)) return rc;
  HX_CHECK(hipMemcpy(pad.data(), src, s->padded * sizeof(float), hipMemcpyDeviceToHost));
  unpack_padded(s, pad, flat);
  return 0;
}
extern "C" int hx_ppo_set_params_h(hx_ppo* s, const float* flat) {
  const int rc = upload_flat(s, s->params, flat);
  s->apack_dirty = true;
  if (rc == 0) refresh_transposes(s, s->stream);
  return rc;
}

extern "C" int hx_ppo_set_compute_dtype(hx_ppo* s, int dtype) {
  if (dtype != 0 && dtype != 1) { hx_set_error("hx_ppo_set_compute_dtype: 0 = f32, 1 = bf16 forward/dgrad"); return -2; }
  if (dtype == 1 && s->frames) { hx_set_error("hx_ppo_set_compute_dtype: the bf16 kernels read ready-made rows; create the learner without frame storage for dtype 1"); return -2; }
  if (dtype == 1 && s->wT[1] == nullptr) {
    for (int net = 0; net < 2; ++net)
      for (int l = 1; l <= 2; ++l) {
        const Layer& L = s->L[net * 4 + l];
        if (L.in_ld % 4 != 0 || L.out % 4 != 0) { hx_set_error("bf16 mode needs hidden widths that are multiples of 4"); return -2; }
        int rc = palloc(s, &s->wT[net * 4 + l], (size_t)L.in_ld * L.out); if (rc) return rc;
      }
  }
  s->bf16 = (dtype == 1);
  refresh_transposes(s, s->stream);
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int hx_ppo_get_params_h(hx_ppo* s, float* flat) { return download_flat(s, s->params, flat); }
extern "C" int hx_ppo_set_opt_state_h(hx_ppo* s, const float* m, const float* v, int64_t step) {
  int rc = upload_flat(s, s->m, m); if (rc) return rc;
  rc = upload_flat(s, s->v, v); if (rc) return rc;
  s->adam_t = step; return 0;
}
extern "C" int hx_ppo_get_opt_state_h(hx_ppo* s, float* m, float* v, int64_t* step) {
  int rc = download_flat(s, s->m, m); if (rc) return rc;
  rc = download_flat(s, s->v, v); if (rc) return rc;
  *step = s->adam_t; return 0;
}

// hidden layers of one network: X[M][ld] -> act[0..2]
static void mlp_hidden_fwd(hx_ppo* s, int net, const float* X, int ldx, int M, float** act, hipStream_t st = nullptr, bool fp32_only = false,
                           const RowTable* rt = nullptr, bool whole_chip = false) {
  const Layer* L = s->L + net * 4;
  const bool bg = (st != nullptr && st == s->stream2) && !whole_chip;      // whole_chip: nothing runs beside it (the flush after the rollout)
  if (!st) st = s->stream;
  gemm_fwd(s, st, X, ldx, s->params + L[0].w, L[0].in_ld, s->params + L[0].b, act[0], M, L[0].out, L[0].in_ld, bg, fp32_only, rt);
  gemm_fwd(s, st, act[0], L[1].in_ld, s->params + L[1].w, L[1].in_ld, s->params + L[1].b, act[1], M, L[1].out, L[1].in_ld, bg, fp32_only);
  gemm_fwd(s, st, act[1], L[2].in_ld, s->params + L[2].w, L[2].in_ld, s->params + L[2].b, act[2], M, L[2].out, L[2].in_ld, bg, fp32_only);
}

// Layers [l0, 3) of BOTH networks at update size, layer l of the actor and of the critic in one launch (hx_gemm_group_kernel):
// fp32, plain rows.  X of a network is only read when l0 == 0.
static void mlp_hidden_fwd_pair(hx_ppo* s, int l0, const float* Xa, int ldxa, const float* Xc, int ldxc, int M, hipStream_t st) {
  for (int l = l0; l < 3; ++l) {
    GemmGroup G{};
    G.n = 2;
    bool kfull = true;
    for (int net = 0; net < 2; ++net) {
      const Layer& Ly = s->L[net * 4 + l];
      float** act = net ? s->act_c : s->act_a;
      GemmArgs& g = G.p[1 - net];      // the critic's (longer K) tiles first in every XCD's queue, so that the launch's tail is short tiles
      g.A = (l == 0) ? (net ? Xc : Xa) : act[l - 1]; g.lda = (l == 0) ? (net ? ldxc : ldxa) : Ly.in_ld;
      g.B = s->params + Ly.w; g.ldb = Ly.in_ld; g.C = act[l]; g.ldc = Ly.out; g.M = M; g.N = Ly.out; g.K = Ly.in_ld; g.bias = s->params + Ly.b;
      kfull = kfull && (g.K % 32 == 0);
    }
    if (kfull && s->gemm_sp && M >= 16384 && G.p[0].K >= 512 && G.p[1].K >= 512) launch_gemm_sp_group<128, 128, 32, true, true, EPI_BIAS_ELU, 2, 2, true>(s, G, st);
    else if (kfull) launch_gemm_group<128, 128, 32, true, true, EPI_BIAS_ELU, true>(s, G, st);
    // the two input layers (K = 616 / 1052): 256 x 256 tiles on eight waves, one workgroup per CU, +1.5 % (profiles/r04_b_gemm_lab.txt, r04_w)
    else if (s->gemm_sp && M >= 16384 && l == 0) launch_gemm_sp_group<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 4, false>(s, G, st);
    else launch_gemm_group<128, 128, 16, true, true, EPI_BIAS_ELU, false>(s, G, st);
  }
}
// input gradients of layer l of both networks in one launch: dZ[l-1] = (dZ[l] W[l]) * elu'(act[l-1])
static void gemm_dgrad_pair(hx_ppo* s, int l, int M, hipStream_t st) {
  GemmGroup G{};
  G.n = 2;
  bool kfull = true;
  for (int net = 0; net < 2; ++net) {
    const Layer& Ly = s->L[net * 4 + l];
    float** act = net ? s->act_c : s->act_a;
    float** dz = net ? s->dz_c : s->dz_a;
    GemmArgs& g = G.p[net];
    g.A = dz[l]; g.lda = Ly.out; g.B = s->params + Ly.w; g.ldb = Ly.in_ld; g.C = dz[l - 1]; g.ldc = Ly.in_ld; g.M = M; g.N = Ly.in_ld; g.K = Ly.out;
    g.H = act[l - 1]; g.ldh = Ly.in_ld;
    kfull = kfull && (g.K % 32 == 0);
  }
  if (kfull && s->gemm_sp && M >= 16384 && G.p[0].K >= 256 && G.p[1].K >= 256) launch_gemm_sp_group<128, 128, 32, true, false, EPI_ELU_GRAD, 2, 2, true>(s, G, st);
  else if (kfull) launch_gemm_group<64, 128, 32, true, false, EPI_ELU_GRAD, true>(s, G, st);
  else launch_gemm_group<64, 128, 32, true, false, EPI_ELU_GRAD, false>(s, G, st);
}

// dynamic LDS of the fused actor: rt = 1: X, H1, H2, H3 side by side; rt = 2: two shared buffers (max(X, H2) and max(H1, H3) wide)
static size_t fa_lds_bytes(int in_ld, int rt) {
  if (rt == 1) return (size_t)(FA_ROWS * (in_ld + 4 + 512 + 4 + 256 + 4 + 128 + 4) + FA_ROWS * MAX_A) * sizeof(float);
  const int wa = (in_ld + 4 > 256 + 4) ? in_ld + 4 : 256 + 4;
  return (size_t)(32 * (wa + 512 + 4)) * sizeof(float);
}
// 16-row workgroups; HX_ACTOR_ROWS=32 selects the 32-row form (measured SLOWER at 4096 rows: collection 18.3 against 16.9 ms per
// iteration, profiles/r04_t_actor_rows.txt -- with 128 workgroups of 8 waves the per-wave MFMA chain, not the L2 stream, sets the time)
static int fa_row_tiles(hx_ppo* s, int rows) {
  (void)rows;
  if (s->actor_rows == 32) return (s->actor_waves == 8 && fa_lds_bytes(s->L[0].in_ld, 2) <= 150 * 1024) ? 2 : 1;
  return 1;
}
// values for rollout slots [crit_done, upto) on the second stream: one critic forward over (slots * N) rows
static int critic_flush(hx_ppo* s, int upto, bool after_rollout = false) {
  const int N = s->cfg.num_envs;
  while (s->crit_done < upto) {
    int slots = upto - s->crit_done;
    const int max_slots = s->Mmax / N > 0 ? s->Mmax / N : 1;
    if (slots > max_slots) slots = max_slots;
    const int rows = slots * N;
    HX_CHECK(hipStreamWaitEvent(s->stream2, s->ev_priv, 0));   // the newest slot's rows have been copied
    if (s->frames) {
      // the gathered first layer on the persistent grid.  Expanding the batch's rows first and running the plain layer was measured
      // and is worse at every grid size of the expansion (profiles/r04_o_frames.txt): a launch beside the fused actor delays it and
      // the critic's own GEMMs more than the gathered loader costs (421 against 344 us per batch)
      const RowTable rt{s->off_priv + (size_t)s->crit_done * N, s->kz_priv + (size_t)s->crit_done * N, s->cfg.num_priv};
      mlp_hidden_fwd(s, 1, s->s_priv, 0, rows, s->act_c, s->stream2, false, &rt, after_rollout);
    } else {
      const float* sp = s->s_priv + (size_t)s->crit_done * N * s->cfg.priv_ld;
      mlp_hidden_fwd(s, 1, sp, s->cfg.priv_ld, rows, s->act_c, s->stream2, false, nullptr, after_rollout);
    }
    hipLaunchKernelGGL(hx_value_head_kernel, dim3((rows + 15) / 16), dim3(256), 0, s->stream2, s->act_c[2], s->cfg.critic_hidden[2],
                       s->params + s->L[7].w, s->params + s->L[7].b, rows, s->s_values + (size_t)s->crit_done * N);
    HX_CHECK(hipGetLastError());
    s->crit_done += slots;
  }
  HX_CHECK(hipEventRecord(s->ev_crit, s->stream2));
  return 0;
}

// PPO.act for the env rows [env0, env0+count) of rollout slot t.  `st` = stream of the caller's env shard;
// with `dual` the critic chain is forked onto the learner's second stream (whole-batch call), otherwise both
// chains run back to back on `st` and the overlap comes from the other shard's stream.
static int act_impl(hx_ppo* s, const float* obs, const float* priv, const float* eps, int env0, int count, hipStream_t st,
                    bool dual, float** actions_out) {
  const int N = s->cfg.num_envs, A = s->cfg.num_actions, t = s->step;
  if (s->frames) { hx_set_error("hx_ppo_act: this learner keeps single-frame observation storage and is driven by hx_rollout; ready-made rows cannot be stored"); return -2; }
  if (t >= s->cfg.num_steps) { hx_set_error("Rollout buffer overflow"); return -10; }   // rollout_storage.py:88-89
  if (env0 < 0 || count <= 0 || env0 + count > N) { hx_set_error("hx_ppo_act: env range out of bounds"); return -2; }
  float* so = s->s_obs + ((size_t)t * N + env0) * s->cfg.obs_ld;
  float* sp = s->s_priv + ((size_t)t * N + env0) * s->cfg.priv_ld;
  // rows already in place when the env step wrote them straight into the storage (hx_sim_step_ex)
  if (obs != so) HX_CHECK(hipMemcpyAsync(so, obs, (size_t)count * s->cfg.obs_ld * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (priv != sp) HX_CHECK(hipMemcpyAsync(sp, priv, (size_t)count * s->cfg.priv_ld * sizeof(float), hipMemcpyDeviceToDevice, st));
  float* aa[3]; float* ac[3];
  for (int l = 0; l < 3; ++l) { aa[l] = s->act_a[l] + (size_t)env0 * s->cfg.actor_hidden[l]; ac[l] = s->act_c[l] + (size_t)env0 * s->cfg.critic_hidden[l]; }
  const int hw = s->cfg.actor_hidden[2], hwc = s->cfg.critic_hidden[2];
  if (t == 0 && env0 == 0) HX_CHECK(hipMemcpyAsync(s->sigma_old, s->params + s->std_off, A * sizeof(float), hipMemcpyDeviceToDevice, st));
  float* acts = s->s_actions + ((size_t)t * N + env0) * A;
  if (dual) {
    // Whole-batch rollout: only the ACTOR is on the critical path (action -> env step -> next observation).
    // The critic's values are first needed by GAE, so the critic runs DEFERRED: every critic_chunk slots one batch
    // (chunk * N rows) on the second, lowest-priority stream, started beside the actor kernel of the step -- the env
    // step, whose waves need a whole SIMD's registers each, is what a resident GEMM wave hurts.
    // slot t's privileged rows are in the storage; the event is only looked at by a flush (an event record costs the
    // stream ~6 us, so not every step)
    const int flush_upto = t + 1;
    const bool flush_now = flush_upto - s->crit_done >= s->critic_chunk;
    if (flush_now && !s->critic_late) HX_CHECK(hipEventRecord(s->ev_priv, st));
    const Layer* La = s->L;
    const bool fused_ok = La[0].out == 512 && La[1].out == 256 && La[2].out == 128 && s->cfg.obs_ld == La[0].in_ld;
    if (fused_ok) {
      if (s->apack_dirty) {
        for (int l = 0; l < 3; ++l) {
          if (!s->apack[l]) { const int rc = palloc(s, &s->apack[l], (size_t)(La[l].out / 16) * ((La[l].in_ld + 15) / 16) * 256); if (rc) return rc; }
          hipLaunchKernelGGL(hx_actor_pack_kernel, dim3(256), dim3(256), 0, st, s->params + La[l].w, La[l].out, La[l].in_ld, La[l].in_ld, s->apack[l]);
        }
        s->apack_dirty = false;
      }
      const int rt = fa_row_tiles(s, count);
      const size_t shm = fa_lds_bytes(La[0].in_ld, rt);
#define HX_FA_ARGS so, s->cfg.obs_ld, count, s->apack[0], s->params + La[0].b, La[0].in_ld, La[0].out, s->apack[1],                        \
                   s->params + La[1].b, La[1].out, s->apack[2], s->params + La[2].b, La[2].out, s->params + La[3].w,                      \
                   s->params + La[3].b, s->params + s->std_off, eps, A, s->seed_lo, s->seed_hi, s->act_counter, s->row_base, acts,        \
                   s->s_mu + ((size_t)t * N + env0) * A, s->s_logp + (size_t)t * N + env0, FrameSrc{nullptr, nullptr, nullptr, 0}, hx_step_book{}, 0, s->pause_flag
      // bf16 mode: the rollout actor rounds its operands exactly like the update's bf16 forward, otherwise the importance
      // ratio of the first epoch is not 1 (with fp32 here training plateaued 36 % lower, profiles/r01_k_bf16.txt)
      const dim3 fgrid((count + 16 * rt - 1) / (16 * rt));
      if (s->actor_waves == 8 && rt == 2) {
        if (s->bf16) hipLaunchKernelGGL((hx_actor_fused_kernel<true, 8, 2>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
        else hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8, 2>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
      } else if (s->actor_waves == 8) {
        if (s->bf16) hipLaunchKernelGGL((hx_actor_fused_kernel<true, 8>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
        else if (s->actor_depth == 3) hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8, 1, 3>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
        else if (s->actor_depth == 4) hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8, 1, 4>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
        else if (s->actor_depth == 6) hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8, 1, 6>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
        else hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
      } else if (s->actor_waves == 2) {
        hipLaunchKernelGGL((hx_actor_fused_kernel<false, 2>), fgrid, dim3(128), shm, st, HX_FA_ARGS);      // probe of a two-wave actor
      } else {
        if (s->bf16) hipLaunchKernelGGL((hx_actor_fused_kernel<true, 4>), fgrid, dim3(256), shm, st, HX_FA_ARGS);
        else hipLaunchKernelGGL((hx_actor_fused_kernel<false, 4>), fgrid, dim3(256), shm, st, HX_FA_ARGS);
      }
#undef HX_FA_ARGS
    } else {
      mlp_hidden_fwd(s, 0, so, s->cfg.obs_ld, count, aa, st);
      hipLaunchKernelGGL(hx_actor_head_kernel, dim3((count + 15) / 16), dim3(256), A * hw * sizeof(float), st, aa[2], hw,
                         s->params + s->L[3].w, s->params + s->L[3].b, s->params + s->std_off, eps, count, A, s->seed_lo, s->seed_hi,
                         s->act_counter, s->row_base, acts, s->s_mu + ((size_t)t * N + env0) * A, s->s_logp + (size_t)t * N + env0);
    }
    if (flush_now) {
      if (s->critic_late) HX_CHECK(hipEventRecord(s->ev_priv, st));      // the burst starts when the actor has finished, beside the env step
      const int rc = critic_flush(s, flush_upto); if (rc) return rc;
    }
  } else {
    mlp_hidden_fwd(s, 0, so, s->cfg.obs_ld, count, aa, st);
    mlp_hidden_fwd(s, 1, sp, s->cfg.priv_ld, count, ac, st);
    hipLaunchKernelGGL(hx_act_head_kernel, dim3((count + 15) / 16), dim3(256), (size_t)(A * hw + hwc) * sizeof(float), st,
                       aa[2], ac[2], hw, hwc, s->params + s->L[3].w, s->params + s->L[3].b, s->params + s->L[7].w, s->params + s->L[7].b,
                       s->params + s->std_off, eps, count, A, s->seed_lo, s->seed_hi + (uint32_t)env0, s->act_counter, s->row_base, acts,
                       s->s_mu + ((size_t)t * N + env0) * A, s->s_values + (size_t)t * N + env0, s->s_logp + (size_t)t * N + env0);
    if (env0 + count == N) s->crit_done = t + 1;             // shard path computes values inline
  }
  HX_CHECK(hipGetLastError());
  if (actions_out) *actions_out = acts;
  return 0;
}

extern "C" int hx_ppo_act(hx_ppo* s, const float* obs, const float* priv, const float* eps, float** actions_out) {
  const int rc = act_impl(s, obs, priv, eps, 0, s->cfg.num_envs, s->stream, true, actions_out);
  s->act_counter++;
  return rc;
}
// the bookkeeping of the env step that produced the current rows, for rollout actors that do not do it themselves
__global__ void __launch_bounds__(256) hx_ppo_book_kernel(hx_step_book b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < b.n) hx_step_book_row(b, e);
  if (blockIdx.x == 0 && threadIdx.x < 64) hx_step_book_global(b, (int)threadIdx.x);
}

// PPO.act for rollout slot s->step of a learner with single-frame storage: the rows of the slot are windows of the frame
// rings (written there by hx_sim_export_stack / hx_sim_step_frames).  `book`: bookkeeping of the env step that completed
// them, still owed (include/hx_sim.h hx_sim_take_book); nullable.
static int act_frames(hx_ppo* s, const hx_step_book* book, float** actions_out) {
  const int N = s->cfg.num_envs, A = s->cfg.num_actions, t = s->step;
  hipStream_t st = s->stream;
  if (t >= s->cfg.num_steps) { hx_set_error("Rollout buffer overflow"); return -10; }   // rollout_storage.py:88-89
  if (t == 0) HX_CHECK(hipMemcpyAsync(s->sigma_old, s->params + s->std_off, A * sizeof(float), hipMemcpyDeviceToDevice, st));
  float* acts = s->s_actions + (size_t)t * N * A;
  const int flush_upto = t + 1;
  const bool flush_now = flush_upto - s->crit_done >= s->critic_chunk;
  if (flush_now && !s->critic_late) HX_CHECK(hipEventRecord(s->ev_priv, st));
  const Layer* La = s->L;
  const bool fused_ok = La[0].out == 512 && La[1].out == 256 && La[2].out == 128;
  const FrameSrc fsrc{s->s_obs, s->off_obs + (size_t)t * N, s->kz_obs + (size_t)t * N, s->cfg.num_obs};
  if (fused_ok) {
    if (s->apack_dirty) {
      for (int l = 0; l < 3; ++l) {
        if (!s->apack[l]) { const int rc = palloc(s, &s->apack[l], (size_t)(La[l].out / 16) * ((La[l].in_ld + 15) / 16) * 256); if (rc) return rc; }
        hipLaunchKernelGGL(hx_actor_pack_kernel, dim3(256), dim3(256), 0, st, s->params + La[l].w, La[l].out, La[l].in_ld, La[l].in_ld, s->apack[l]);
      }
      s->apack_dirty = false;
    }
    const int rt = fa_row_tiles(s, N);
    const size_t shm = fa_lds_bytes(La[0].in_ld, rt);
    hx_step_book bk{}; if (book) bk = *book;
#define HX_FA_ARGS (const float*)nullptr, 0, N, s->apack[0], s->params + La[0].b, La[0].in_ld, La[0].out, s->apack[1],                       \
                   s->params + La[1].b, La[1].out, s->apack[2], s->params + La[2].b, La[2].out, s->params + La[3].w,                      \
                   s->params + La[3].b, s->params + s->std_off, (const float*)nullptr, A, s->seed_lo, s->seed_hi, s->act_counter, s->row_base, acts, \
                   s->s_mu + (size_t)t * N * A, s->s_logp + (size_t)t * N, fsrc, bk, book ? 1 : 0, s->pause_flag
    const dim3 fgrid((N + 16 * rt - 1) / (16 * rt));
    if (s->actor_waves == 8 && rt == 2) hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8, 2>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
    else if (s->actor_waves == 8) hipLaunchKernelGGL((hx_actor_fused_kernel<false, 8>), fgrid, dim3(512), shm, st, HX_FA_ARGS);
    else hipLaunchKernelGGL((hx_actor_fused_kernel<false, 4>), fgrid, dim3(256), shm, st, HX_FA_ARGS);
#undef HX_FA_ARGS
  } else {
    if (book) hipLaunchKernelGGL(hx_ppo_book_kernel, dim3((N + 255) / 256), dim3(256), 0, st, *book);
    const RowTable rt{fsrc.off, fsrc.kz, fsrc.klim};
    mlp_hidden_fwd(s, 0, s->s_obs, 0, N, s->act_a, st, false, &rt);
    const int hw = s->cfg.actor_hidden[2];
    hipLaunchKernelGGL(hx_actor_head_kernel, dim3((N + 15) / 16), dim3(256), A * hw * sizeof(float), st, s->act_a[2], hw,
                       s->params + s->L[3].w, s->params + s->L[3].b, s->params + s->std_off, (const float*)nullptr, N, A, s->seed_lo, s->seed_hi,
                       s->act_counter, s->row_base, acts, s->s_mu + (size_t)t * N * A, s->s_logp + (size_t)t * N);
  }
  if (flush_now) {
    if (s->critic_late) HX_CHECK(hipEventRecord(s->ev_priv, st));
    const int rc = critic_flush(s, flush_upto); if (rc) return rc;
  }
  HX_CHECK(hipGetLastError());
  if (actions_out) *actions_out = acts;
  return 0;
}
extern "C" int hx_ppo_act_range(hx_ppo* s, const float* obs, const float* priv, const float* eps, int env0, int count, void* stream,
                                float** actions_out) {
  return act_impl(s, obs, priv, eps, env0, count, (hipStream_t)stream, false, actions_out);
}

static int process_impl(hx_ppo* s, const float* rew, const uint8_t* dones, const uint8_t* timeouts, int env0, int count, hipStream_t st) {
  const int N = s->cfg.num_envs, t = s->step;
  if (s->frames) { hx_set_error("hx_ppo_process_step: this learner keeps single-frame observation storage and is driven by hx_rollout"); return -2; }
  if (t >= s->cfg.num_steps) { hx_set_error("Rollout buffer overflow"); return -10; }
  hipLaunchKernelGGL(hx_process_step_kernel, dim3((count + 255) / 256), dim3(256), 0, st, rew, dones, timeouts, count,
                     s->s_rewards + (size_t)t * N + env0, s->s_dones + (size_t)t * N + env0, s->s_timeouts + (size_t)t * N + env0);
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int hx_ppo_process_step(hx_ppo* s, const float* rew, const uint8_t* dones, const uint8_t* timeouts) {
  const int rc = process_impl(s, rew, dones, timeouts, 0, s->cfg.num_envs, s->stream);
  if (rc == 0) s->step += 1;
  return rc;
}
// shard form: `advance` != 0 on the last shard of a step moves the rollout slot forward
extern "C" int hx_ppo_process_step_range(hx_ppo* s, const float* rew, const uint8_t* dones, const uint8_t* timeouts, int env0, int count,
                                         void* stream, int advance) {
  const int rc = process_impl(s, rew, dones, timeouts, env0, count, (hipStream_t)stream);
  if (rc == 0 && advance) { s->step += 1; s->act_counter++; }
  return rc;
}

// bootstrap values V(s_T) for env rows [env0, env0+count)  (first half of PPO.compute_returns, ppo.py:116)
extern "C" int hx_ppo_last_values_range(hx_ppo* s, const float* last_priv, int env0, int count, void* stream) {
  hipStream_t st = stream ? (hipStream_t)stream : s->stream;
  // activations in the backward pass's scratch (idle until the update): the deferred critic's last batch, which owns act_c, may still
  // be running on the second stream while this runs (hx_ppo_compute_returns)
  float* ac[3];
  for (int l = 0; l < 3; ++l) ac[l] = s->dz_c[l] + (size_t)env0 * s->cfg.critic_hidden[l];
  mlp_hidden_fwd(s, 1, last_priv, s->cfg.priv_ld, count, ac, st);
  hipLaunchKernelGGL(hx_value_head_kernel, dim3((count + 15) / 16), dim3(256), 0, st, ac[2], s->cfg.critic_hidden[2],
                     s->params + s->L[7].w, s->params + s->L[7].b, count, s->last_values + env0);
  HX_CHECK(hipGetLastError());
  return 0;
}

// last_priv == NULL: the bootstrap values were already produced by hx_ppo_last_values_range
extern "C" int hx_ppo_compute_returns(hx_ppo* s, const float* last_priv) {
  const int N = s->cfg.num_envs, T = s->cfg.num_steps;
  // finish the deferred critic for every stored slot, then make the main stream wait for it
  // (whatever the background critic did not get to -- it yields to the actor -- runs here on the whole chip, not on its half-chip grid)
  if (s->crit_done < s->step) { HX_CHECK(hipEventRecord(s->ev_priv, s->stream)); const int rc = critic_flush(s, s->step, true); if (rc) return rc; }
  // the bootstrap values need nothing of the deferred critic: they run while its last batch finishes on the second stream (130 us of
  // the ~300 us between the rollout's last launch and GAE, profiles/r04_ba_critic_tiles.txt)
  if (last_priv) { const int rc = hx_ppo_last_values_range(s, last_priv, 0, N, s->stream); if (rc) return rc; }
  HX_CHECK(hipStreamWaitEvent(s->stream, s->ev_crit, 0));
  HX_CHECK(hipMemsetAsync(s->moments, 0, 3 * sizeof(double), s->stream));
  hipLaunchKernelGGL(hx_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, s->stream, s->s_rewards, s->s_dones, s->s_timeouts, s->s_values, s->last_values,
                     T, N, s->cfg.gamma, s->cfg.lam, s->s_returns, s->s_adv_raw, s->moments);
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int hx_ppo_adv_moments(hx_ppo* s, void** m) { *m = s->moments; return 0; }
extern "C" int hx_ppo_adv_normalize(hx_ppo* s) {
  const size_t TN = (size_t)s->cfg.num_steps * s->cfg.num_envs;
  if (s->comm) { const int rc = hx_comm_all_reduce(s->comm, s->moments, 3, HX_COMM_F64, HX_COMM_SUM, s->stream); if (rc) return rc; }
  hipLaunchKernelGGL(hx_adv_normalize_kernel, dim3((unsigned)((TN + 255) / 256)), dim3(256), 0, s->stream, s->s_adv_raw, s->moments, TN, s->s_adv);
  HX_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hx_ppo_update_begin(hx_ppo* s, const int32_t* perm) {
  const int TN = s->cfg.num_steps * s->cfg.num_envs;
  if (perm) HX_CHECK(hipMemcpyAsync(s->perm, perm, (size_t)TN * sizeof(int), hipMemcpyDeviceToDevice, s->stream));
  else hipLaunchKernelGGL(hx_perm_kernel, dim3((TN + 255) / 256), dim3(256), 0, s->stream, s->perm, TN, s->perm_key + 0x9E3779B9u * (s->perm_counter++));
  if (s->frames) {
    MbTableArgs g{};
    g.perm = s->perm; g.TN = TN;
    g.off_obs = s->off_obs; g.off_priv = s->off_priv; g.kz_obs = s->kz_obs; g.kz_priv = s->kz_priv;
    g.mb_off_obs = s->mb_off_obs; g.mb_off_priv = s->mb_off_priv; g.mb_kz_obs = s->mb_kz_obs; g.mb_kz_priv = s->mb_kz_priv;
    g.actions = s->s_actions; g.mu = s->s_mu; g.values = s->s_values; g.returns = s->s_returns; g.logp = s->s_logp; g.adv = s->s_adv;
    g.A = s->cfg.num_actions; g.row_mb = s->row_mb_all;
    hipLaunchKernelGGL(hx_mb_tables_kernel, dim3((unsigned)(((long long)TN * (2 * g.A + 8) + 255) / 256)), dim3(256), 0, s->stream, g);
    HX_CHECK(hipGetLastError());
  }
  SchedState z{};
  // keep lr, clear the loss accumulators
  HX_CHECK(hipMemsetAsync(&s->sched->vloss_sum, 0, 2 * sizeof(float), s->stream));
  s->mb_done = 0;
  s->mb_gathered = 0;
  s->mb_total = s->cfg.num_learning_epochs * s->cfg.num_mini_batches;
  (void)z;
  return 0;
}

extern "C" int hx_ppo_minibatch_backward(hx_ppo* s, int mb_index, void** grad_buffer, int64_t* count) {
  const hx_ppo_cfg& c = s->cfg;
  const int A = c.num_actions;
  const int TN = c.num_steps * c.num_envs;
  const int M = TN / c.num_mini_batches;
  const int mb = mb_index % c.num_mini_batches;       // the permutation is reused by every epoch (rollout_storage.py:149,165)
  hipStream_t st = s->stream;
  const int slot = (s->mb_slots > 1) ? mb : 0;
  RowTable rt_obs{}, rt_priv{};
  const bool gathered_loaders = s->frames && s->frames_gather;
  if (gathered_loaders) {
    // HX_FRAMES_GATHER=1 (round 3): the first-layer forward products read the minibatch's rows in place through the permutation-
    // ordered tables of hx_ppo_update_begin and, as a side effect, leave the assembled rows in a one-minibatch workspace (obs_mb /
    // priv_mb), from which the first-layer weight-gradient products read them with the plain loader
    s->obs_mb = s->obs_mb_all; s->priv_mb = s->priv_mb_all;
    s->row_mb = s->row_mb_all + (size_t)mb * M * (2 * A + 4);
    rt_obs = RowTable{s->mb_off_obs + (size_t)mb * M, s->mb_kz_obs + (size_t)mb * M, c.num_obs, s->obs_mb, c.obs_ld};
    rt_priv = RowTable{s->mb_off_priv + (size_t)mb * M, s->mb_kz_priv + (size_t)mb * M, c.num_priv, s->priv_mb, c.priv_ld};
  } else {
    s->obs_mb = s->obs_mb_all + (size_t)slot * s->Mmax * c.obs_ld;
    s->priv_mb = s->priv_mb_all + (size_t)slot * s->Mmax * c.priv_ld;
    s->row_mb = s->frames ? s->row_mb_all + (size_t)mb * M * (2 * A + 4) : s->row_mb_all + (size_t)slot * s->Mmax * (2 * A + 4);
  }
  if (s->frames && !gathered_loaders && (s->mb_slots == 1 || !((s->mb_gathered >> mb) & 1ull))) {
    // frame storage: the minibatch's rows expanded from the rings, once per update (the epochs share the permutation)
    ExpandMbArgs ea{s->s_obs, s->s_priv, s->mb_off_obs + (size_t)mb * M, s->mb_kz_obs + (size_t)mb * M, s->mb_off_priv + (size_t)mb * M, s->mb_kz_priv + (size_t)mb * M,
                    c.num_obs, c.num_priv, c.obs_ld, c.priv_ld, s->obs_mb, s->priv_mb};
    hipLaunchKernelGGL(hx_expand_mb_kernel, dim3(M), dim3(256), 0, st, ea);
    s->mb_gathered |= (1ull << mb);
  }
  if (!s->frames && (s->mb_slots == 1 || !((s->mb_gathered >> mb) & 1ull))) {
    GatherArgs ga{};
    ga.idx = s->perm + (size_t)mb * M; ga.M = M;
    ga.obs = s->s_obs; ga.obs_ld = c.obs_ld; ga.obs_mb = s->obs_mb;
    ga.priv = s->s_priv; ga.priv_ld = c.priv_ld; ga.priv_mb = s->priv_mb;
    ga.actions = s->s_actions; ga.mu = s->s_mu; ga.values = s->s_values; ga.returns = s->s_returns; ga.logp = s->s_logp; ga.adv = s->s_adv;
    ga.A = A; ga.row_mb = s->row_mb;
    hipLaunchKernelGGL(hx_gather_kernel, dim3(M), dim3(256), 0, st, ga);
    s->mb_gathered |= (1ull << mb);
  }
  // forward.  The actor's and the critic's chains are independent up to the loss head: they run on two streams, so that the
  // short launches of one (240 .. 480 workgroups on 768 slots) fill up beside the other's and no launch waits for the
  // previous one's tail; the critic joins before the loss head and forks again for the backward pass.
  hipStream_t sb = s->stream_b ? s->stream_b : st;
  if (s->stream_b) { HX_CHECK(hipEventRecord(s->ev_b0, st)); HX_CHECK(hipStreamWaitEvent(sb, s->ev_b0, 0)); }
  // HX_GEMM_PAIR: layer l of both networks in one launch (fp32, update size, one stream); with single-frame storage the two
  // gathered input layers keep their own launches
  const bool pair = s->gemm_pair && !s->bf16 && !s->stream_b && M >= 16384 && s->fwd_in_tile == 128;
  if (pair && !gathered_loaders) mlp_hidden_fwd_pair(s, 0, s->obs_mb, c.obs_ld, s->priv_mb, c.priv_ld, M, st);
  else if (pair) {
    const Layer* La = s->L; const Layer* Lc = s->L + 4;
    gemm_fwd(s, st, s->s_obs, c.obs_ld, s->params + La[0].w, La[0].in_ld, s->params + La[0].b, s->act_a[0], M, La[0].out, La[0].in_ld, false, false, &rt_obs);
    gemm_fwd(s, st, s->s_priv, c.priv_ld, s->params + Lc[0].w, Lc[0].in_ld, s->params + Lc[0].b, s->act_c[0], M, Lc[0].out, Lc[0].in_ld, false, false, &rt_priv);
    mlp_hidden_fwd_pair(s, 1, nullptr, 0, nullptr, 0, M, st);
  } else {
    mlp_hidden_fwd(s, 0, gathered_loaders ? s->s_obs : s->obs_mb, c.obs_ld, M, s->act_a, st, false, gathered_loaders ? &rt_obs : nullptr);
    mlp_hidden_fwd(s, 1, gathered_loaders ? s->s_priv : s->priv_mb, c.priv_ld, M, s->act_c, sb, false, gathered_loaders ? &rt_priv : nullptr);
  }
  if (s->stream_b) { HX_CHECK(hipEventRecord(s->ev_b1, sb)); HX_CHECK(hipStreamWaitEvent(st, s->ev_b1, 0)); }
  // heads: losses + gradient into the third hidden layer
  const int hw = c.actor_hidden[2], hwc = c.critic_hidden[2];
  const int hblocks = (M + HEAD_ROWS - 1) / HEAD_ROWS;
  HeadArgs h{};
  h.h3a = s->act_a[2]; h.h3c = s->act_c[2]; h.hw = hw; h.hwc = hwc;
  h.W4 = s->params + s->L[3].w; h.b4 = s->params + s->L[3].b; h.W4c = s->params + s->L[7].w; h.b4c = s->params + s->L[7].b;
  h.stdp = s->params + s->std_off; h.sigma_old = s->sigma_old; h.row_mb = s->row_mb; h.M = M; h.A = A;
  h.clip = c.clip_param; h.vcoef = c.value_loss_coef; h.ecoef = c.entropy_coef; h.use_clipped_value_loss = c.use_clipped_value_loss;
  h.dz3a = s->dz_a[2]; h.dz3c = s->dz_c[2]; h.slab = s->head_slab; h.slab_w = s->head_slab_w;
  h.stage_c = head_lds_bytes(hw, hwc, A, true) <= 64 * 1024;
  const size_t shm = head_lds_bytes(hw, hwc, A, h.stage_c != 0);
  // hector-shaped heads on the matrix cores (hx_loss_head_mfma_kernel); HX_HEAD_MFMA=0 keeps the VALU kernel
  const bool head_mfma = s->head_mfma && A <= HEADM_NA && hw % 64 == 0 && hwc % 64 == 0 && head_mfma_lds_bytes(hw, hwc) <= 64 * 1024;
  if (head_mfma) hipLaunchKernelGGL(hx_loss_head_mfma_kernel, dim3(hblocks), dim3(256), head_mfma_lds_bytes(hw, hwc), st, h);
  else if (A <= 16) hipLaunchKernelGGL(hx_loss_head_kernel<16>, dim3(hblocks), dim3(256), shm, st, h);
  else hipLaunchKernelGGL(hx_loss_head_kernel<32>, dim3(hblocks), dim3(256), shm, st, h);
  HeadScatter hs{s->L[3].w, s->L[3].b, s->L[7].w, s->L[7].b, s->std_off, s->stats_off, A, hw, hwc};
  const int hchunk = 32, hchunks = (hblocks + hchunk - 1) / hchunk;
  hipLaunchKernelGGL(hx_slab_chunk_kernel, dim3((s->head_slab_w + 255) / 256, hchunks), dim3(256), 0, st, s->head_slab, hblocks, s->head_slab_w, hchunk, s->head_slab2);
  hipLaunchKernelGGL(hx_head_scatter_kernel, dim3((s->head_slab_w + 255) / 256), dim3(256), 0, st, s->head_slab2, hchunks, s->head_slab_w, s->grads, hs, (float)M);
  // backward through the hidden layers of both networks; partial slabs go to per-layer regions, one reduce at the end
  ReduceTable rt{}; unsigned blocks = 0;
  // Grouped weight gradients (hx_gemm_group_kernel): the input-gradient chain of both networks first, then every layer's
  // dZ^T X in one or two launches.  fp32 path with whole K tiles only; HX_WGRAD_GROUP=0 launches layer by layer as before.
  const bool multi = s->wgrad_multi && !s->bf16 && (M % 32 == 0) && M >= 512;
  const bool grouped = !multi && s->wgrad_group && !s->bf16 && (M % HX_BK_UPD == 0);
  WgradJob jobs[8]; int njobs = 0;
  WgradLayerDesc wl[6]; WgradOperands wo[6];
  if (s->stream_b) { HX_CHECK(hipEventRecord(s->ev_b0, st)); HX_CHECK(hipStreamWaitEvent(sb, s->ev_b0, 0)); }
  for (int l = 2; l >= 0; --l) {
    for (int net = 0; net < 2; ++net) {
      const hipStream_t st = net ? sb : s->stream;          // shadows: the critic's backward chain on the second stream
      const Layer* L = s->L + net * 4;
      float** act = net ? s->act_c : s->act_a;
      float** dz = net ? s->dz_c : s->dz_a;
      const float* X = net ? s->priv_mb : s->obs_mb;
      const int ldx = net ? c.priv_ld : c.obs_ld;
      const float* in = (l == 0) ? X : act[l - 1];
      const int ld_in = (l == 0) ? ldx : L[l].in_ld;
      if (multi) {
        wl[net * 3 + l] = WgradLayerDesc{L[l].out, L[l].in_ld};
        wo[net * 3 + l] = WgradOperands{dz[l], in, ld_in};
      } else {
        float* slab = s->slab + s->slab_off[net * 4 + l];
        float* bslab = s->bias_slab + s->bslab_off[net * 4 + l];
        int bparts = 0, splits = 0;
        if (grouped) jobs[njobs++] = WgradJob{dz[l], L[l].out, in, ld_in, L[l].in_ld, slab, bslab, s->slab_splits[net * 4 + l], rt.nseg};
        else splits = gemm_wgrad(s, st, dz[l], L[l].out, in, ld_in, L[l].in_ld, M, slab, bslab, &bparts, s->slab_splits[net * 4 + l]);
        const unsigned cnt = (unsigned)L[l].out * (unsigned)L[l].in_ld;
        int k = rt.nseg;
        rt.src[k] = slab; rt.dst[k] = s->grads + L[l].w; rt.count[k] = cnt; rt.cols[k] = cnt; rt.ldd[k] = cnt; rt.S[k] = splits; rt.block0[k] = blocks; blocks += (cnt + 1023) / 1024;
        k = ++rt.nseg;
        rt.src[k] = bslab; rt.dst[k] = s->grads + L[l].b; rt.count[k] = (unsigned)L[l].out; rt.cols[k] = rt.count[k]; rt.ldd[k] = rt.count[k]; rt.S[k] = bparts; rt.block0[k] = blocks; blocks += (L[l].out + 1023) / 1024;
        ++rt.nseg;
      }
      if (l > 0 && !pair) gemm_dgrad(s, st, dz[l], L[l].out, s->params + L[l].w, L[l].in_ld, act[l - 1], dz[l - 1], M, L[l].in_ld, L[l].out, s->wT[net * 4 + l]);
    }
    if (l > 0 && pair) gemm_dgrad_pair(s, l, M, st);
  }
  if (s->stream_b) { HX_CHECK(hipEventRecord(s->ev_b1, sb)); HX_CHECK(hipStreamWaitEvent(st, s->ev_b1, 0)); }
  if (grouped) {
    int sp[8], bp[8];
    gemm_wgrad_groups(s, st, jobs, njobs, M, sp, bp);
    for (int i = 0; i < njobs; ++i) { rt.S[jobs[i].seg] = sp[i]; rt.S[jobs[i].seg + 1] = bp[i]; }
  }
  if (multi) {
    // every layer's dZ^T X at one workgroup per CU (hx_wgrad_multi_kernel): plan per minibatch size, made on first use
    const WgradPlan* plan = nullptr;
    if (int rc = wgrad_plan_for(s, wl, M, &plan)) return rc;
    static const int kid = prof_register("hx_wgrad_multi_kernel");
    for (int q = 0; q < plan->nlaunch; ++q) {
      WgradMulti W;
      hx_wgrad_fill(*plan, q, wl, wo, M, s->wslab, s->wbslab, W);
      double flops = 0.0;
      for (const WgradPiece& pc : plan->pieces) if (pc.launch == q) flops += 2.0 * wl[pc.layer].out * pc.ncols * (double)M;
      ProfScope ps(s, kid, st, flops);
      hipLaunchKernelGGL(hx_wgrad_multi_kernel, dim3(hx_group_grid(W.G)), dim3(256), 0, st, W);
    }
    for (const WgradPiece& pc : plan->pieces) {
      const Layer& Ly = s->L[(pc.layer / 3) * 4 + pc.layer % 3];
      int k = rt.nseg++;
      rt.src[k] = s->wslab + pc.slab_off; rt.dst[k] = s->grads + Ly.w + pc.col0; rt.count[k] = (unsigned)Ly.out * (unsigned)pc.ncols;
      rt.cols[k] = (unsigned)pc.ncols; rt.ldd[k] = (unsigned)Ly.in_ld; rt.S[k] = pc.splits; rt.block0[k] = blocks; blocks += (rt.count[k] + 1023) / 1024;
      if (pc.bias) {
        k = rt.nseg++;
        rt.src[k] = s->wbslab + pc.bslab_off; rt.dst[k] = s->grads + Ly.b; rt.count[k] = (unsigned)Ly.out; rt.cols[k] = rt.count[k]; rt.ldd[k] = rt.count[k];
        rt.S[k] = pc.splits * pc.tiles_n; rt.block0[k] = blocks; blocks += (Ly.out + 1023) / 1024;
      }
    }
  }
  rt.block0[rt.nseg] = blocks;
  hipLaunchKernelGGL(hx_reduce_all_kernel, dim3(blocks), dim3(256), 0, st, rt);
  HX_CHECK(hipGetLastError());
  if (grad_buffer) *grad_buffer = s->grads;
  if (count) *count = (int64_t)s->padded + 4;
  return 0;
}

extern "C" int hx_ppo_minibatch_step(hx_ppo* s, float inv_world) {
  const hx_ppo_cfg& c = s->cfg;
  hipStream_t st = s->stream;
  if (s->comm) {
    // the one collective of an optimiser step: gradients and loss statistics of all ranks, in place, on this stream
    const int rc = hx_comm_all_reduce(s->comm, s->grads, s->padded + 4, HX_COMM_F32, HX_COMM_SUM, st); if (rc) return rc;
    inv_world = 1.0f / (float)hx_comm_world(s->comm);
  }
  // after an all-reduce(sum) the statistics are global sums, so kl_mean = kl_sum / rows is already world-wide
  hipLaunchKernelGGL(hx_schedule_kernel, dim3(1), dim3(1), 0, st, s->grads + s->stats_off, c.adaptive_schedule, c.desired_kl, s->sched, s->sumsq);
  hipLaunchKernelGGL(hx_sumsq_kernel, dim3(256), dim3(256), 0, st, s->grads, s->padded, inv_world, s->sumsq);
  s->adam_t += 1;
  const float bc1 = 1.0f - (float)pow(0.9, (double)s->adam_t);
  const float bc2s = (float)sqrt(1.0 - pow(0.999, (double)s->adam_t));
  hipLaunchKernelGGL(hx_adam_kernel, dim3((unsigned)((s->padded + 255) / 256)), dim3(256), 0, st, s->params, s->grads, s->m, s->v, s->padded,
                     inv_world, s->sumsq, c.max_grad_norm, s->sched, bc1, bc2s);
  s->apack_dirty = true;
  refresh_transposes(s, st);
  HX_CHECK(hipGetLastError());
  s->mb_done += 1;
  return 0;
}

extern "C" int hx_ppo_update_end(hx_ppo* s, float* stats_h) {
  SchedState st;
  HX_CHECK(hipMemcpyAsync(&st, s->sched, sizeof(st), hipMemcpyDeviceToHost, s->stream));
  // the stream carries this update's all-reduces when data parallel: wait with the communicator's deadline, not forever
  if (s->comm) { const int rc = hx_comm_wait(s->comm, s->stream, 0.0); if (rc) return rc; }
  else HX_CHECK(hipStreamSynchronize(s->stream));
  const float n = (float)(s->mb_done > 0 ? s->mb_done : 1);
  if (stats_h) { stats_h[0] = st.vloss_sum / n; stats_h[1] = st.sloss_sum / n; stats_h[2] = st.lr; stats_h[3] = st.last_kl; }
  s->step = 0;       // storage.clear(), ppo.py:182
  s->crit_done = 0;
  return 0;
}

extern "C" int hx_ppo_update(hx_ppo* s, const int32_t* perm, float* stats_h) {
  int rc = hx_ppo_update_begin(s, perm); if (rc) return rc;
  for (int i = 0; i < s->mb_total; ++i) {
    rc = hx_ppo_minibatch_backward(s, i, nullptr, nullptr); if (rc) return rc;
    rc = hx_ppo_minibatch_step(s, 1.0f); if (rc) return rc;
  }
  return hx_ppo_update_end(s, stats_h);
}

extern "C" int hx_ppo_buffer(hx_ppo* s, int which, void** d) {
  switch (which) {
    case HX_PPO_BUF_ACTIONS: *d = s->s_actions; break;
    case HX_PPO_BUF_VALUES: *d = s->s_values; break;
    case HX_PPO_BUF_LOGP: *d = s->s_logp; break;
    case HX_PPO_BUF_MU: *d = s->s_mu; break;
    case HX_PPO_BUF_REWARDS: *d = s->s_rewards; break;
    case HX_PPO_BUF_RETURNS: *d = s->s_returns; break;
    case HX_PPO_BUF_ADVANTAGES: *d = s->s_adv; break;
    case HX_PPO_BUF_GRADS: *d = s->grads; break;
    case HX_PPO_BUF_PERM: *d = s->perm; break;
    case HX_PPO_BUF_OBS: case HX_PPO_BUF_PRIV:
      if (s->frames) { hx_set_error("hx_ppo_buffer: with single-frame storage the observation rows do not exist in memory: hx_ppo_storage_rows expands them"); return -2; }
      *d = (which == HX_PPO_BUF_OBS) ? s->s_obs : s->s_priv; break;
    case HX_PPO_BUF_DONES: *d = s->s_dones; break;
    case HX_PPO_BUF_TIMEOUTS: *d = s->s_timeouts; break;
    default: hx_set_error("hx_ppo_buffer: unknown id"); return -2;
  }
  return 0;
}
extern "C" int hx_ppo_storage_rows(hx_ppo* s, int which, int t0, int t1, float* dst) {
  if (!s || !dst || (which != HX_PPO_BUF_OBS && which != HX_PPO_BUF_PRIV)) { hx_set_error("hx_ppo_storage_rows: bad argument"); return -2; }
  const int N = s->cfg.num_envs, T = s->cfg.num_steps;
  const bool ob = which == HX_PPO_BUF_OBS;
  const int ld = ob ? s->cfg.obs_ld : s->cfg.priv_ld;
  if (t0 < 0 || t1 <= t0 || t1 > T + (s->frames ? 1 : 0)) { hx_set_error("hx_ppo_storage_rows: slot range out of bounds"); return -2; }
  if (!s->frames) {
    HX_CHECK(hipMemcpyAsync(dst, (ob ? s->s_obs : s->s_priv) + (size_t)t0 * N * ld, (size_t)(t1 - t0) * N * ld * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    return 0;
  }
  hipLaunchKernelGGL(hx_expand_rows_kernel, dim3((unsigned)((t1 - t0) * N)), dim3(256), 0, s->stream, ob ? s->s_obs : s->s_priv,
                     (ob ? s->off_obs : s->off_priv) + (size_t)t0 * N, (ob ? s->kz_obs : s->kz_priv) + (size_t)t0 * N, ob ? s->cfg.num_obs : s->cfg.num_priv, ld, dst, (t1 - t0) * N);
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int hx_ppo_get_lr(hx_ppo* s, float* lr) {
  SchedState st;
  HX_CHECK(hipMemcpyAsync(&st, s->sched, sizeof(st), hipMemcpyDeviceToHost, s->stream));
  if (s->comm) { const int rc = hx_comm_wait(s->comm, s->stream, 0.0); if (rc) return rc; }
  else HX_CHECK(hipStreamSynchronize(s->stream));
  *lr = st.lr; return 0;
}
extern "C" int hx_ppo_set_lr(hx_ppo* s, float lr) {
  HX_CHECK(hipMemcpyAsync(&s->sched->lr, &lr, sizeof(float), hipMemcpyHostToDevice, s->stream));
  if (int rc = learner_sync(s)) return rc;
  return 0;
}

extern "C" int hx_ppo_inference(hx_ppo* s, const float* obs, int rows, float* out) {
  if (rows > s->Mmax) { hx_set_error("hx_ppo_inference: rows > workspace"); return -2; }
  mlp_hidden_fwd(s, 0, obs, s->cfg.obs_ld, rows, s->act_a);
  const int A = s->cfg.num_actions;
  hipLaunchKernelGGL(hx_mean_head_kernel, dim3((rows * A + 255) / 256), dim3(256), 0, s->stream, s->act_a[2], s->cfg.actor_hidden[2],
                     s->params + s->L[3].w, s->params + s->L[3].b, rows, A, out);
  HX_CHECK(hipGetLastError());
  return 0;
}

// launches of the deferred critic on the background stream are not bracketed: they share the chip with the rollout's
// kernels, so an event pair there measures contended time, not the kernel
ProfScope::ProfScope(hx_ppo* s_, int kid, hipStream_t st_, double flops) : s(s_), st(st_), on(false) {
  if (!s || !s->prof || (s->prof_only >= 0 && s->prof_only != kid) || st == s->stream2) return;
  // an event pair costs the stream ~7 us around the launch (kernel trace: 0 us between unbracketed launches): in the timed
  // region of a benchmark only every prof_every-th launch of the selected symbol is bracketed -- a uniform sample
  if (s->prof_every > 1 && (s->prof_seen++ % s->prof_every) != 0) return;
  while (s->ev_used + 2 > s->ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; s->ev.push_back(e); s->ev_kid.push_back(0); }
  if ((size_t)kid >= s->prof_flops.size()) { s->prof_flops.resize(kid + 1, 0.0); s->prof_launches.resize(kid + 1, 0); }
  (void)hipEventRecord(s->ev[s->ev_used], st);
  s->ev_kid[s->ev_used] = kid;
  s->prof_flops[kid] += flops; s->prof_launches[kid] += 1;
  on = true;
}
ProfScope::~ProfScope() { if (on) { (void)hipEventRecord(s->ev[s->ev_used + 1], st); s->ev_used += 2; } }

extern "C" int hx_ppo_prof_begin(hx_ppo* s, const char* only_symbol, int sample_every) {
  if (!s) { hx_set_error("hx_ppo_prof_begin: null learner"); return -2; }
  s->prof_only = -1;
  s->prof_every = sample_every > 1 ? sample_every : 1; s->prof_seen = 0;
  if (only_symbol && *only_symbol) {
    const auto& names = prof_names();
    for (size_t i = 0; i < names.size(); ++i) if (names[i] == only_symbol) s->prof_only = (int)i;
    if (s->prof_only < 0) { hx_set_error(std::string("hx_ppo_prof_begin: no launch of symbol '") + only_symbol + "' has been seen yet"); return -2; }
  }
  s->ev_used = 0; s->prof = true;
  s->prof_flops.assign(prof_names().size(), 0.0); s->prof_launches.assign(prof_names().size(), 0);
  return 0;
}
extern "C" int hx_ppo_prof_end(hx_ppo* s, hx_prof_row* rows, int max_rows, int* n_rows) {
  if (!s || !n_rows) { hx_set_error("hx_ppo_prof_end: null argument"); return -2; }
  s->prof = false;
  if (int rc = learner_sync(s)) return rc;
  std::vector<double> ms(prof_names().size(), 0.0);
  for (size_t i = 0; i + 1 < s->ev_used; i += 2) { float t = 0; HX_CHECK(hipEventElapsedTime(&t, s->ev[i], s->ev[i + 1])); if ((size_t)s->ev_kid[i] < ms.size()) ms[s->ev_kid[i]] += t; }
  int n = 0;
  for (size_t k = 0; k < s->prof_launches.size() && k < ms.size(); ++k) {
    if (s->prof_launches[k] == 0) continue;
    if (rows && n < max_rows) {
      hx_prof_row& r = rows[n];
      memset(&r, 0, sizeof(r));
      strncpy(r.symbol, prof_names()[k].c_str(), sizeof(r.symbol) - 1);
      r.ms = ms[k]; r.launches = s->prof_launches[k]; r.flops = s->prof_flops[k];
    }
    ++n;
  }
  *n_rows = n;
  s->ev_used = 0;
  return 0;
}

// The rollout loop of OnPolicyRunner.learn (on_policy_runner.py:127-138) for `steps` env steps, driven from C so
// that launch issue (~40 launches per step with two shards) never waits for the Python interpreter.
// One shard: whole-batch calls (critic chain forked on the learner's second stream).  Several shards: every
// shard advances on its own stream; one shard's env-step kernel overlaps the other shards' GEMMs.
// frame-mode rollout: 2 launches per step -- the actor reads its rows as windows of the frame rings (and does the previous
// step's bookkeeping), the env step writes the new frames, their rows' first-valid-element entries and nothing else.
static int rollout_frames(hx_ppo* p, hx_sim* sim, int steps) {
  const int N = p->cfg.num_envs, T = p->cfg.num_steps;
  if (hx_sim_stream(sim) != (void*)p->stream) { hx_set_error("hx_rollout: simulator and learner must share one stream"); return -2; }
  const int slot0 = p->step;
  auto slot_of = [&](int frame_obs, int frame_priv, int row) {
    hx_frame_slot f{};
    f.obs = p->s_obs + (size_t)frame_obs * p->fo; f.obs_env_stride = (int64_t)p->Po * p->fo;
    f.priv = p->s_priv + (size_t)frame_priv * p->fp; f.priv_env_stride = (int64_t)p->Pp * p->fp;
    f.obs_kz = p->kz_obs + (size_t)row * N; f.priv_kz = p->kz_priv + (size_t)row * N;
    return f;
  };
  if (slot0 == 0) {
    // the simulator's current stack becomes frames 0 .. stack-1 of every ring (row 0)
    const hx_frame_slot f0 = slot_of(0, 0, 0);
    const int rc = hx_sim_export_stack(sim, &f0); if (rc) return rc;
  } else if (p->frames_sim_step != hx_sim_step_counter(sim)) {
    hx_set_error("hx_rollout: the simulator stepped outside this learner's rollout since slot 0 (frame rings are stale)"); return -2;
  }
  for (int t = 0; t < steps; ++t) {
    const int slot = p->step;
    if (slot >= T) { hx_set_error("Rollout buffer overflow"); return -10; }
    hx_step_book book{}; int32_t owed = 0;
    int rc = hx_sim_take_book(sim, &book, &owed); if (rc) return rc;
    float* act = nullptr;
    rc = act_frames(p, owed ? &book : nullptr, &act); if (rc) return rc;
    p->act_counter++;
    // row slot+1 = the windows ending at the frame this step produces: frame index slot + stack of either stream
    const hx_frame_slot f = slot_of(slot + p->So, slot + p->Sp, slot + 1);
    rc = hx_sim_step_frames(sim, act, nullptr, &f, p->s_rewards + (size_t)slot * N, p->s_dones + (size_t)slot * N, p->s_timeouts + (size_t)slot * N);
    if (rc) return rc;
    p->step += 1;
  }
  // the last step's bookkeeping, and the simulator's own row buffers back in step with the rings (HX_BUF_OBS / HX_BUF_PRIV are
  // what the caller bootstraps from and what a row-API step would shift)
  int rc = hx_sim_flush_book(sim); if (rc) return rc;
  const hx_frame_slot fl = slot_of(p->step, p->step, p->step);
  rc = hx_sim_import_stack(sim, &fl); if (rc) return rc;
  p->frames_sim_step = hx_sim_step_counter(sim);
  return 0;
}

extern "C" int hx_rollout(hx_ppo* p, hx_sim** sims, const int32_t* env0, const int32_t* count, int nshards, int steps) {
  const int N = p->cfg.num_envs, T = p->cfg.num_steps;
  // the count of fused-actor workgroups in flight starts every rollout at zero: a decrement lost to an aborted launch cannot
  // outlive the iteration (the background critic's bounded wait, hx_pause_poll, would otherwise be paid again every flush)
  if (p->pause_flag && p->step == 0) HX_CHECK(hipMemsetAsync(p->pause_flag, 0, 2 * sizeof(int), p->stream));
  // single-shard row rollouts: the critic also sleeps through the stacking launch (HX_STACK_PAUSE)
  // (only with the fused rollout actor: it is that kernel's first workgroup that takes the stacking launch's count back)
  const bool fused_actor = p->L[0].out == 512 && p->L[1].out == 256 && p->L[2].out == 128 && p->cfg.obs_ld == p->L[0].in_ld;
  const bool stack_pause = p->pause_flag && p->stack_pause && fused_actor && !p->frames && nshards == 1 && hx_sim_stream(sims[0]) == (void*)p->stream;
  struct PauseWordGuard { hx_sim* s; ~PauseWordGuard() { if (s) (void)hx_sim_set_pause_word(s, nullptr); } } pause_guard{nullptr};      // off again on every way out
  if (stack_pause) { const int rc = hx_sim_set_pause_word(sims[0], p->pause_flag); if (rc) return rc; pause_guard.s = sims[0]; }
  if (p->frames) {
    if (nshards != 1 || count[0] != N) { hx_set_error("hx_rollout: single-frame storage takes one simulator with all of the learner's robots"); return -2; }
    return rollout_frames(p, sims[0], steps);
  }
  for (int t = 0; t < steps; ++t) {
    for (int h = 0; h < nshards; ++h) {
      void *obs, *priv, *rew, *rst, *tov;
      int rc = hx_sim_buffer(sims[h], HX_BUF_OBS, &obs); if (rc) return rc;
      rc = hx_sim_buffer(sims[h], HX_BUF_PRIV, &priv); if (rc) return rc;
      float* act = nullptr;
      if (nshards == 1) {
        // whole batch, zero-copy: the env writes observation t+1, reward, done and time-out of step t straight into
        // the rollout storage (slot t+1 rows / slot t scalars); 3 launches per step: actor, env step, stack
        const int slot = p->step;
        if (slot >= T) { hx_set_error("Rollout buffer overflow"); return -10; }
        rc = hx_ppo_act(p, (const float*)obs, (const float*)priv, nullptr, &act); if (rc) return rc;
        float* od = nullptr; float* pd = nullptr;
        if (slot + 1 < T) { od = p->s_obs + (size_t)(slot + 1) * N * p->cfg.obs_ld; pd = p->s_priv + (size_t)(slot + 1) * N * p->cfg.priv_ld; }
        rc = hx_sim_step_ex(sims[h], act, nullptr, od, pd, p->s_rewards + (size_t)slot * N, p->s_dones + (size_t)slot * N,
                            p->s_timeouts + (size_t)slot * N);
        if (rc) return rc;
        p->step += 1;
      } else {
        rc = hx_ppo_act_range(p, (const float*)obs, (const float*)priv, nullptr, env0[h], count[h], hx_sim_stream(sims[h]), &act);
        if (rc) return rc;
        rc = hx_sim_step(sims[h], act, nullptr); if (rc) return rc;
        hx_sim_buffer(sims[h], HX_BUF_REW, &rew); hx_sim_buffer(sims[h], HX_BUF_RESET, &rst); hx_sim_buffer(sims[h], HX_BUF_TIMEOUT_VISIBLE, &tov);
        rc = hx_ppo_process_step_range(p, (const float*)rew, (const uint8_t*)rst, (const uint8_t*)tov, env0[h], count[h],
                                       hx_sim_stream(sims[h]), h == nshards - 1);
        if (rc) return rc;
      }
    }
  }
  if (stack_pause) HX_CHECK(hipMemsetAsync(p->pause_flag, 0, 2 * sizeof(int), p->stream));      // the last stacking launch's count has no actor behind it
  return 0;
}

// ---- device memory helpers
extern "C" int hx_malloc(size_t bytes, void** out) { HX_CHECK(hipMalloc(out, bytes)); HX_CHECK(hipMemset(*out, 0, bytes)); return 0; }
extern "C" int hx_free(void* p) { HX_CHECK(hipFree(p)); return 0; }
extern "C" int hx_memcpy_h2d(void* d, const void* s, size_t b, void* st) {
  if (st) HX_CHECK(hipStreamSynchronize((hipStream_t)st));
  HX_CHECK(hipMemcpy(d, s, b, hipMemcpyHostToDevice)); return 0;
}
extern "C" int hx_memcpy_d2h(void* d, const void* s, size_t b, void* st) {
  if (st) HX_CHECK(hipStreamSynchronize((hipStream_t)st));
  HX_CHECK(hipMemcpy(d, s, b, hipMemcpyDeviceToHost)); return 0;
}
extern "C" int hx_memcpy_d2d(void* d, const void* s, size_t b, void* st) { HX_CHECK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, (hipStream_t)st)); return 0; }
extern "C" int hx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int hx_set_device(int i) { HX_CHECK(hipSetDevice(i)); return 0; }
