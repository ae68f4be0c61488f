// ORACLE-SIDE TOOL (test infrastructure, never shipped) -- compiled host restatement of the learner's dense arithmetic.
//
// The learner's kernels (isaac_amd/csrc/hx_gemm.h, hx_ppo.hip) are MFMA code and have no host form, so unlike the env step
// (hx_host.cpp compiles the kernels' own text) the CPU side of the learner is a restatement: the three dense products of an
// MLP layer (reference: nn.Linear + nn.ELU in humanoid/algo/ppo/actor_critic.py:57-80 and their autograd, as written out in
// oracle/ppo.py MLP.forward / MLP.backward), clip_grad_norm_ + Adam (ppo.py:171-174), on all host cores:
//   forward   H[M,N]  = elu(X[M,K] W[N,K]^T + b)
//   dgrad     dX[M,K] = (dZ[M,N] W[N,K]) * elu'(H_prev)
//   wgrad     dW[N,K] = dZ[M,N]^T X[M,K] ;  db[N] = column sums of dZ
// One fp32 GEMM (Goto / BLIS structure: B block packed once per K slice and shared, one 6 x KC A panel per thread in L1,
// 6 x 16 AVX2-FMA micro-kernel) serves all three through operand strides.  oracle/host/learner.py wraps it as a drop-in for
// oracle.ppo.MLP, so the numpy oracle's PPO (loss head, GAE, schedule) runs unchanged on top of it; tests/test_host_learner.py
// pins it against the reference-generated fixture tests/golden/ppo_small.npz and against numpy.  Used by bench.py's
// cpu_baseline leg (SURVEY.md 8d ii) -- never by the product.
#include <immintrin.h>
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr int MR = 6, NR = 16, KC = 192;

// C[6][16] (+)= Ap[kc][6] * Bp[kc][16]
inline void micro_6x16(int kc, const float* __restrict__ Ap, const float* __restrict__ Bp, float* __restrict__ C, int ldc, bool acc, int mr, int nr) {
  __m256 c[MR][2];
  for (int i = 0; i < MR; ++i) { c[i][0] = _mm256_setzero_ps(); c[i][1] = _mm256_setzero_ps(); }
  for (int k = 0; k < kc; ++k) {
    const __m256 b0 = _mm256_load_ps(Bp + k * NR), b1 = _mm256_load_ps(Bp + k * NR + 8);
    const float* a = Ap + k * MR;
#pragma GCC unroll 6
    for (int i = 0; i < MR; ++i) {
      const __m256 ai = _mm256_broadcast_ss(a + i);
      c[i][0] = _mm256_fmadd_ps(ai, b0, c[i][0]);
      c[i][1] = _mm256_fmadd_ps(ai, b1, c[i][1]);
    }
  }
  if (mr == MR && nr == NR) {
    for (int i = 0; i < MR; ++i) {
      float* d = C + (size_t)i * ldc;
      if (acc) { c[i][0] = _mm256_add_ps(c[i][0], _mm256_loadu_ps(d)); c[i][1] = _mm256_add_ps(c[i][1], _mm256_loadu_ps(d + 8)); }
      _mm256_storeu_ps(d, c[i][0]); _mm256_storeu_ps(d + 8, c[i][1]);
    }
  } else {
    alignas(32) float t[MR][NR];
    for (int i = 0; i < MR; ++i) { _mm256_store_ps(t[i], c[i][0]); _mm256_store_ps(t[i] + 8, c[i][1]); }
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < nr; ++j) { float* d = C + (size_t)i * ldc + j; *d = acc ? *d + t[i][j] : t[i][j]; }
  }
}

// C[M][N] = A(M x K) B(K x N):  A(i,k) = A[i*rsa + k*csa],  B(k,j) = B[k*rsb + j*csb]
void sgemm(int M, int N, int K, const float* A, int64_t rsa, int64_t csa, const float* B, int64_t rsb, int64_t csb, float* C, int ldc) {
  const int npan = (N + NR - 1) / NR;
  float* Bp = (float*)aligned_alloc(64, (size_t)npan * KC * NR * sizeof(float));
  const int mpan = (M + MR - 1) / MR;
#pragma omp parallel
  {
    float* Ap = (float*)aligned_alloc(64, (size_t)KC * MR * sizeof(float));
    for (int pc = 0; pc < K; pc += KC) {
      const int kc = std::min(KC, K - pc);
      // the K slice of B, packed once as NR-wide panels [panel][k][16] (zero padded), by all threads
#pragma omp for schedule(static)
      for (int jp = 0; jp < npan; ++jp) {
        float* d = Bp + (size_t)jp * KC * NR;
        const int j0 = jp * NR, nr = std::min(NR, N - j0);
        for (int k = 0; k < kc; ++k) {
          const float* s = B + (size_t)(pc + k) * rsb + (size_t)j0 * csb;
          if (csb == 1 && nr == NR) memcpy(d + k * NR, s, NR * sizeof(float));
          else { for (int j = 0; j < nr; ++j) d[k * NR + j] = s[(size_t)j * csb]; for (int j = nr; j < NR; ++j) d[k * NR + j] = 0.f; }
        }
      }   // implicit barrier
#pragma omp for schedule(static)
      for (int ip = 0; ip < mpan; ++ip) {
        const int i0 = ip * MR, mr = std::min(MR, M - i0);
        for (int k = 0; k < kc; ++k)
          for (int i = 0; i < MR; ++i) Ap[k * MR + i] = (i < mr) ? A[(size_t)(i0 + i) * rsa + (size_t)(pc + k) * csa] : 0.f;
        for (int jp = 0; jp < npan; ++jp)
          micro_6x16(kc, Ap, Bp + (size_t)jp * KC * NR, C + (size_t)i0 * ldc + jp * NR, ldc, pc > 0, mr, std::min(NR, N - jp * NR));
      }   // implicit barrier: Bp is rewritten by the next slice
    }
    free(Ap);
  }
  free(Bp);
}
// exp on 8 lanes (Cephes expf: range reduction by ln 2, degree-5 polynomial; relative error ~1e-7 on the range used here)
inline __m256 exp256(__m256 x) {
  x = _mm256_max_ps(_mm256_min_ps(x, _mm256_set1_ps(88.f)), _mm256_set1_ps(-88.f));
  __m256 fx = _mm256_floor_ps(_mm256_fmadd_ps(x, _mm256_set1_ps(1.44269504088896341f), _mm256_set1_ps(0.5f)));
  x = _mm256_fnmadd_ps(fx, _mm256_set1_ps(0.693359375f), x);
  x = _mm256_fnmadd_ps(fx, _mm256_set1_ps(-2.12194440e-4f), x);
  const __m256 z = _mm256_mul_ps(x, x);
  __m256 y = _mm256_set1_ps(1.9875691500e-4f);
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.3981999507e-3f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(8.3334519073e-3f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(4.1665795894e-2f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.6666665459e-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(5.0000001201e-1f));
  y = _mm256_add_ps(_mm256_fmadd_ps(y, z, x), _mm256_set1_ps(1.f));
  const __m256i e = _mm256_slli_epi32(_mm256_add_epi32(_mm256_cvttps_epi32(fx), _mm256_set1_epi32(127)), 23);
  return _mm256_mul_ps(y, _mm256_castsi256_ps(e));
}
// h[j] = elu(h[j] + b[j]) over a row; ELU(alpha = 1) = z > 0 ? z : exp(z) - 1 (the product's hx_elu: exp through the hardware exp2)
inline void bias_elu_row(float* h, const float* b, int n) {
  int j = 0;
  const __m256 one = _mm256_set1_ps(1.f), zero = _mm256_setzero_ps();
  for (; j + 8 <= n; j += 8) {
    const __m256 z = _mm256_add_ps(_mm256_loadu_ps(h + j), _mm256_loadu_ps(b + j));
    const __m256 neg = _mm256_sub_ps(exp256(_mm256_min_ps(z, zero)), one);
    _mm256_storeu_ps(h + j, _mm256_blendv_ps(neg, z, _mm256_cmp_ps(z, zero, _CMP_GT_OQ)));
  }
  for (; j < n; ++j) { const float z = h[j] + b[j]; h[j] = z > 0.f ? z : expm1f(z); }
}
}  // namespace

extern "C" {
int hxl_num_threads(void) { return omp_get_max_threads(); }

// H[M][N] = act(X[M][K] W[N][K]^T + b);  act = ELU if `activate` else identity
void hxl_linear_forward(int M, int N, int K, const float* X, const float* W, const float* b, float* H, int activate) {
  sgemm(M, N, K, X, K, 1, W, 1, K, H, N);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    float* h = H + (size_t)i * N;
    if (activate) bias_elu_row(h, b, N);
    else for (int j = 0; j < N; ++j) h[j] += b[j];
  }
}
// dW[N][K] = dZ[M][N]^T X[M][K];  db[N] = sum over rows of dZ
void hxl_linear_wgrad(int M, int N, int K, const float* dZ, const float* X, float* dW, float* db) {
  sgemm(N, K, M, dZ, 1, N, X, K, 1, dW, K);
  // column sums: every thread sums a block of rows (contiguous reads), the partial rows are added in thread order
  const int nt = omp_get_max_threads();
  std::vector<float> part((size_t)nt * N, 0.f);
#pragma omp parallel
  {
    float* p = part.data() + (size_t)omp_get_thread_num() * N;
#pragma omp for schedule(static)
    for (int i = 0; i < M; ++i) { const float* r = dZ + (size_t)i * N; for (int j = 0; j < N; ++j) p[j] += r[j]; }
  }
  for (int j = 0; j < N; ++j) { float s = 0.f; for (int t = 0; t < nt; ++t) s += part[(size_t)t * N + j]; db[j] = s; }
}
// dZprev[M][K] = (dZ[M][N] W[N][K]) * elu'(Hprev[M][K])     (elu'(h) = 1 for h > 0 else h + 1, in terms of the activation)
void hxl_linear_dgrad(int M, int N, int K, const float* dZ, const float* W, const float* Hprev, float* dZprev) {
  sgemm(M, K, N, dZ, N, 1, W, K, 1, dZprev, K);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    float* d = dZprev + (size_t)i * K;
    const float* h = Hprev + (size_t)i * K;
    for (int j = 0; j < K; ++j) d[j] *= (h[j] > 0.f ? 1.f : h[j] + 1.f);
  }
}
// sum of squares in double (clip_grad_norm_, ppo.py:173)
double hxl_sumsq(const float* g, int64_t n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += (double)g[i] * (double)g[i];
  return s;
}
// one tensor of torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8) on the clipped gradient g * coef, in place
void hxl_adam(float* p, const float* g, float* m, float* v, int64_t n, float coef, float lr_over_bc1, float sqrt_bc2) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g[i] * coef;
    const float mi = 0.9f * m[i] + (1.0f - 0.9f) * gi;
    const float vi = 0.999f * v[i] + (1.0f - 0.999f) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_over_bc1 * (mi / (std::sqrt(vi) / sqrt_bc2 + 1e-8f));
  }
}
}
