// gemm_lab: the learner's GEMM shapes (61 440-row minibatch of the hector networks) through the production kernels of hx_gemm.h
// and the slot-placed kernels of hx_gemm_sp.h, same inputs, outputs compared bit for bit, interleaved timing rounds in ONE process
// (cdna_hip_programming.md 5.4 rule 24).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_lab.hip -o tools/micro/gemm_lab
// usage: gemm_lab [rounds=5] [launches per round=4] [case filter substring]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include "../../isaac_amd/csrc/hx_wgrad_plan.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static int rup(int a, int b) { return (a + b - 1) / b * b; }

static float* dev_random(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

struct Variant { std::string name; void (*launch)(GemmGroup&, hipStream_t); int bm, bn; int slots = 768; };
static const int MAXSPLITS = 48;

template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, bool KFULL> static void launch_old(GemmGroup& G, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < G.n; ++i) { GemmArgs& g = G.p[i]; g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN; G.first[i] = blocks; blocks += g.tiles_m * g.tiles_n * (EPI == EPI_SLAB ? g.splits : 1); }
  G.first[G.n] = blocks;
  hipLaunchKernelGGL((hx_gemm_group_kernel<BM, BN, BKT, AK, BK_, EPI, KFULL>), dim3(hx_group_grid(G)), dim3(256), 0, st, G);
}
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, int WM, int WN, bool KFULL> static void launch_sp(GemmGroup& G, hipStream_t st) {
  int blocks = 0;
  for (int i = 0; i < G.n; ++i) { GemmArgs& g = G.p[i]; g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN; G.first[i] = blocks; blocks += g.tiles_m * g.tiles_n * (EPI == EPI_SLAB ? g.splits : 1); }
  G.first[G.n] = blocks;
  hipLaunchKernelGGL((hx_gemm_sp_group_kernel<BM, BN, BKT, AK, BK_, EPI, WM, WN, KFULL>), dim3(hx_group_grid(G)), dim3(64 * WM * WN), 0, st, G);
}

#define WGRAD_EXTRA \
    c.v.push_back({"sp  128x128x32 4w 512", launch_sp<128, 128, 32, false, false, EPI_SLAB, 2, 2, true>, 128, 128, 512}); \
    c.v.push_back({"sp  256x128x16 4w 512", launch_sp<256, 128, 16, false, false, EPI_SLAB, 2, 2, true>, 256, 128, 512}); \
    c.v.push_back({"sp  128x256x16 4w 512", launch_sp<128, 256, 16, false, false, EPI_SLAB, 2, 2, true>, 128, 256, 512}); \
    c.v.push_back({"sp  256x128x16 8w 512", launch_sp<256, 128, 16, false, false, EPI_SLAB, 4, 2, true>, 256, 128, 512}); \
    c.v.push_back({"sp  256x256x16 4w 256", launch_sp<256, 256, 16, false, false, EPI_SLAB, 2, 2, true>, 256, 256, 256}); \
    c.v.push_back({"sp  256x256x16 8w 256", launch_sp<256, 256, 16, false, false, EPI_SLAB, 2, 4, true>, 256, 256, 256});
// stream-K launch of a pair (hx_gemm_sk_kernel): workspace shared by all cases
static float* g_sk_partial = nullptr; static unsigned* g_sk_flags = nullptr; static int* g_sk_err = nullptr; static unsigned g_sk_epoch = 0; static int g_sk_wgs = 256;
template <int BM, int BN, int BKT, bool AK, bool BK_, int EPI, int WM, int WN, bool KFULL> static void launch_sk(GemmGroup& G, hipStream_t st) {
  GemmSk S{};
  S.G = G;
  long long it = 0; int maxnk = 0;
  for (int i = 0; i < G.n; ++i) {
    GemmArgs& g = S.G.p[i]; g.tiles_m = (g.M + BM - 1) / BM; g.tiles_n = (g.N + BN - 1) / BN;
    S.nk[i] = (g.K + BKT - 1) / BKT; S.it0[i] = it; it += (long long)g.tiles_m * g.tiles_n * S.nk[i]; maxnk = std::max(maxnk, S.nk[i]);
  }
  S.it0[G.n] = it;
  if (it / g_sk_wgs < maxnk) { fprintf(stderr, "sk: range shorter than a tile's K loop\n"); exit(1); }
  if (!g_sk_partial) {
    CK(hipMalloc(&g_sk_partial, (size_t)(g_sk_wgs + 1) * 256 * 256 * 4)); CK(hipMalloc(&g_sk_flags, (g_sk_wgs + 1) * 4)); CK(hipMalloc(&g_sk_err, 4));
    CK(hipMemset(g_sk_flags, 0, (g_sk_wgs + 1) * 4)); CK(hipMemset(g_sk_err, 0, 4));
  }
  S.partial = g_sk_partial; S.flags = g_sk_flags; S.err = g_sk_err; S.epoch = ++g_sk_epoch;
  hipLaunchKernelGGL((hx_gemm_sk_kernel<BM, BN, BKT, AK, BK_, EPI, WM, WN, KFULL>), dim3(g_sk_wgs), dim3(64 * WM * WN), 0, st, S);
}
struct Case {
  std::string name; GemmGroup G; double flops; std::vector<Variant> v;
  std::vector<std::pair<float*, size_t>> outs;     // output buffers (compared between variants)
};

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5, per = argc > 2 ? atoi(argv[2]) : 4;
  const char* filter = argc > 3 ? argv[3] : "";
  const int ROWS = getenv("LAB_ROWS") ? atoi(getenv("LAB_ROWS")) : 61440;
  hipStream_t st; CK(hipStreamCreate(&st));
  // operands: activations / pre-activation gradients of both networks
  const int a_in = 616, a_h[3] = {512, 256, 128}, c_in = 1052, c_h[3] = {768, 256, 128};
  float* Xa = dev_random((size_t)ROWS * a_in, 1); float* Xc = dev_random((size_t)ROWS * c_in, 2);
  float *Ha[3], *Hc[3], *Za[3], *Zc[3], *Wa[3], *Wc[3], *ba[3], *bc[3];
  for (int l = 0; l < 3; ++l) {
    Ha[l] = dev_random((size_t)ROWS * a_h[l], 10 + l); Hc[l] = dev_random((size_t)ROWS * c_h[l], 20 + l);
    Za[l] = dev_random((size_t)ROWS * a_h[l], 30 + l); Zc[l] = dev_random((size_t)ROWS * c_h[l], 40 + l);
    Wa[l] = dev_random((size_t)a_h[l] * (l ? a_h[l - 1] : a_in), 50 + l); Wc[l] = dev_random((size_t)c_h[l] * (l ? c_h[l - 1] : c_in), 60 + l);
    ba[l] = dev_random(a_h[l], 70 + l); bc[l] = dev_random(c_h[l], 80 + l);
  }
  std::vector<Case> cases;
  auto outbuf = [&](Case& c, size_t n) { float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4)); c.outs.push_back({d, n}); return d; };

  // ---- weight gradients: the two production groups (768 slots: 3 workgroups per CU)
  auto wgrad_case = [&](const char* name, std::vector<int> members, int target) {
    // members: 0..2 actor layers, 3..5 critic layers
    Case c; c.name = name; c.G = GemmGroup{}; c.flops = 0;
    int T = 0;
    for (int m : members) { const int out = m < 3 ? a_h[m] : c_h[m - 3]; const int in = m < 3 ? (m ? a_h[m - 1] : a_in) : (m > 3 ? c_h[m - 4] : c_in); T += ((out + 127) / 128) * ((in + 127) / 128); }
    int splits = std::max(1, target / T);
    const int kchunk = rup((ROWS + splits - 1) / splits, 32); splits = (ROWS + kchunk - 1) / kchunk;
    c.G.n = (int)members.size();
    for (size_t i = 0; i < members.size(); ++i) {
      const int m = members[i]; const bool actor = m < 3; const int l = actor ? m : m - 3;
      const int out = actor ? a_h[l] : c_h[l]; const int in = actor ? (l ? a_h[l - 1] : a_in) : (l ? c_h[l - 1] : c_in);
      GemmArgs& g = c.G.p[i];
      g.A = actor ? Za[l] : Zc[l]; g.lda = out; g.B = l ? (actor ? Ha[l - 1] : Hc[l - 1]) : (actor ? Xa : Xc); g.ldb = in;
      g.M = out; g.N = in; g.K = ROWS; g.ldc = in; g.splits = splits; g.kchunk = kchunk; g.db_parts = (in + 127) / 128;
      g.C = outbuf(c, (size_t)MAXSPLITS * out * in); g.dbias = outbuf(c, (size_t)MAXSPLITS * 16 * out);
      c.flops += 2.0 * out * in * ROWS;
    }
    printf("# %s: %d tiles x %d slices, kchunk %d\n", name, T, splits, kchunk);
    return c;
  };
  {
    Case c = wgrad_case("wgrad_critic_in", {3}, 768);
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, false, false, EPI_SLAB, true>, 128, 128});
    c.v.push_back({"sp  128x128x16 4w", launch_sp<128, 128, 16, false, false, EPI_SLAB, 2, 2, true>, 128, 128});
    WGRAD_EXTRA
    cases.push_back(c);
  }
  {
    Case c = wgrad_case("wgrad_other5", {0, 4, 1, 2, 5}, 768);
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, false, false, EPI_SLAB, true>, 128, 128});
    c.v.push_back({"sp  128x128x16 4w", launch_sp<128, 128, 16, false, false, EPI_SLAB, 2, 2, true>, 128, 128});
    WGRAD_EXTRA
    cases.push_back(c);
  }
  // ---- single products for calibrating the planner's shape penalties: one shape each, one workgroup per CU
  {
    Case c = wgrad_case("cal_512x616", {0}, 768);
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, false, false, EPI_SLAB, true>, 128, 128});
    c.v.push_back({"sp  512x128x16 s1 256", launch_sp<512, 128, 16, false, false, EPI_SLAB, 4, 1, true>, 512, 128, 256});
    c.v.push_back({"sp  256x256x16 s0 256", launch_sp<256, 256, 16, false, false, EPI_SLAB, 2, 2, true>, 256, 256, 256});
    cases.push_back(c);
  }
  {
    Case c = wgrad_case("cal_256x768", {4}, 768);
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, false, false, EPI_SLAB, true>, 128, 128});
    c.v.push_back({"sp  256x256x16 s0 256", launch_sp<256, 256, 16, false, false, EPI_SLAB, 2, 2, true>, 256, 256, 256});
    c.v.push_back({"sp  128x256x16 s2 256", launch_sp<128, 256, 16, false, false, EPI_SLAB, 1, 4, true>, 128, 256, 256});
    c.v.push_back({"sp  128x128x16 s3 256", launch_sp<128, 128, 16, false, false, EPI_SLAB, 2, 2, true>, 128, 128, 256});
    cases.push_back(c);
  }
  {
    Case c = wgrad_case("cal_strip768x28", {3}, 768);
    GemmArgs& g = c.G.p[0]; g.B = Xc + 1024; g.N = 28; g.ldc = 28; g.db_parts = 1; g.dbias = nullptr;
    c.flops = 2.0 * 768 * 28 * ROWS;
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, false, false, EPI_SLAB, true>, 128, 128});
    c.v.push_back({"sp  512x32x16 s4 16", launch_sp<512, 32, 16, false, false, EPI_SLAB, 4, 1, true>, 512, 32, 16});
    c.v.push_back({"sp  512x32x16 s4 32", launch_sp<512, 32, 16, false, false, EPI_SLAB, 4, 1, true>, 512, 32, 32});
    c.v.push_back({"sp  128x128x16 s3 24", launch_sp<128, 128, 16, false, false, EPI_SLAB, 2, 2, true>, 128, 128, 24});
    cases.push_back(c);
  }
  // ---- forward pairs (critic member first, as in production)
  auto fwd_case = [&](const char* name, int l) {
    Case c; c.name = name; c.G = GemmGroup{}; c.G.n = 2; c.flops = 0;
    for (int net = 0; net < 2; ++net) {
      const bool actor = (net == 1);
      const int out = actor ? a_h[l] : c_h[l]; const int in = actor ? (l ? a_h[l - 1] : a_in) : (l ? c_h[l - 1] : c_in);
      GemmArgs& g = c.G.p[net];
      g.A = l ? (actor ? Ha[l - 1] : Hc[l - 1]) : (actor ? Xa : Xc); g.lda = in; g.B = actor ? Wa[l] : Wc[l]; g.ldb = in;
      g.M = ROWS; g.N = out; g.K = in; g.ldc = out; g.bias = actor ? ba[l] : bc[l]; g.C = outbuf(c, (size_t)ROWS * out);
      c.flops += 2.0 * ROWS * out * in;
    }
    return c;
  };
  {
    Case c = fwd_case("fwd_in_pair", 0);
    c.v.push_back({"old 128x128x16 4w", launch_old<128, 128, 16, true, true, EPI_BIAS_ELU, false>, 128, 128});
    c.v.push_back({"sp  128x128x16 4w", launch_sp<128, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 128, 128});
    c.v.push_back({"sp  256x128x16 4w", launch_sp<256, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 256, 128});
    c.v.push_back({"sp  256x256x16 4w", launch_sp<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 256, 256});
    c.v.push_back({"sp  256x256x16 8w", launch_sp<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 4, false>, 256, 256});
    c.v.push_back({"sk  256x256x16 4w", launch_sk<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 256, 256});
    c.v.push_back({"sk  256x128x16 4w", launch_sk<256, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 256, 128});
    c.v.push_back({"sk  128x256x16 4w", launch_sk<128, 256, 16, true, true, EPI_BIAS_ELU, 2, 2, false>, 128, 256});
    cases.push_back(c);
  }
  for (int l = 1; l < 3; ++l) {
    Case c = fwd_case(l == 1 ? "fwd_h1_pair" : "fwd_h2_pair", l);
    c.v.push_back({"old 128x128x32 4w", launch_old<128, 128, 32, true, true, EPI_BIAS_ELU, true>, 128, 128});
    c.v.push_back({"sp  128x128x32 4w", launch_sp<128, 128, 32, true, true, EPI_BIAS_ELU, 2, 2, true>, 128, 128});
    c.v.push_back({"sp  128x128x16 4w", launch_sp<128, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, true>, 128, 128});
    c.v.push_back({"sp  256x128x16 4w", launch_sp<256, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, true>, 256, 128});
    c.v.push_back({"sp  256x256x16 4w", launch_sp<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 2, true>, 256, 256});
    c.v.push_back({"sk  256x256x16 4w", launch_sk<256, 256, 16, true, true, EPI_BIAS_ELU, 2, 2, true>, 256, 256});
    c.v.push_back({"sk  256x128x16 4w", launch_sk<256, 128, 16, true, true, EPI_BIAS_ELU, 2, 2, true>, 256, 128});
    cases.push_back(c);
  }
  // ---- input-gradient pairs: dX[rows][in] = dZ[rows][out] W[out][in] * elu'(H_prev)
  for (int l = 2; l >= 1; --l) {
    Case c; c.name = l == 2 ? "dgrad_h2_pair" : "dgrad_h1_pair"; c.G = GemmGroup{}; c.G.n = 2; c.flops = 0;
    for (int net = 0; net < 2; ++net) {
      const bool actor = (net == 0);
      const int out = actor ? a_h[l] : c_h[l]; const int in = actor ? a_h[l - 1] : c_h[l - 1];
      GemmArgs& g = c.G.p[net];
      g.A = actor ? Za[l] : Zc[l]; g.lda = out; g.B = actor ? Wa[l] : Wc[l]; g.ldb = in; g.M = ROWS; g.N = in; g.K = out; g.ldc = in;
      g.H = actor ? Ha[l - 1] : Hc[l - 1]; g.ldh = in; g.C = outbuf(c, (size_t)ROWS * in);
      c.flops += 2.0 * ROWS * out * in;
    }
    c.v.push_back({"old  64x128x32 4w", launch_old<64, 128, 32, true, false, EPI_ELU_GRAD, true>, 64, 128});
    c.v.push_back({"sp  128x128x32 4w", launch_sp<128, 128, 32, true, false, EPI_ELU_GRAD, 2, 2, true>, 128, 128});
    c.v.push_back({"sp  128x128x16 4w", launch_sp<128, 128, 16, true, false, EPI_ELU_GRAD, 2, 2, true>, 128, 128});
    c.v.push_back({"sp  256x128x16 4w", launch_sp<256, 128, 16, true, false, EPI_ELU_GRAD, 2, 2, true>, 256, 128});
    c.v.push_back({"sp  256x256x16 4w", launch_sp<256, 256, 16, true, false, EPI_ELU_GRAD, 2, 2, true>, 256, 256});
    c.v.push_back({"sk  256x256x16 4w", launch_sk<256, 256, 16, true, false, EPI_ELU_GRAD, 2, 2, true>, 256, 256});
    cases.push_back(c);
  }

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // ---- all six weight gradients: the production's two grouped launches against the planned one-workgroup-per-CU launches
  if (!filter[0] || std::string("wgrad_all").find(filter) != std::string::npos) {
    const int order[6] = {3, 0, 4, 1, 2, 5};      // layer ids as in wgrad_case: 0..2 actor, 3..5 critic
    WgradLayerDesc LD[6]; WgradOperands OP[6];
    for (int i = 0; i < 6; ++i) {
      const int m = order[i]; const bool actor = m < 3; const int l = actor ? m : m - 3;
      LD[i].out = actor ? a_h[l] : c_h[l]; LD[i].in_ld = actor ? (l ? a_h[l - 1] : a_in) : (l ? c_h[l - 1] : c_in);
      OP[i].dZ = actor ? Za[l] : Zc[l]; OP[i].X = l ? (actor ? Ha[l - 1] : Hc[l - 1]) : (actor ? Xa : Xc); OP[i].ldx = LD[i].in_ld;
    }
    int cus = 256; { int dev = 0; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)); }
    const int slots = getenv("LAB_SLOTS") ? atoi(getenv("LAB_SLOTS")) : cus;
    WgradPlan plan = hx_wgrad_plan(LD, 6, ROWS, slots);
    printf("# wgrad_all plan: %d launches, slab %.1f MB, estimated makespan %.0f unit-rows (ideal %.0f)\n", plan.nlaunch, plan.slab_floats * 4e-6, plan.est_unit_rows,
           1496.0 * ROWS / (4.0 * slots));
    for (const WgradPiece& p : plan.pieces)
      printf("#   layer %d (%d x %d) cols [%d, %d) shape %d (%dx%d) tiles %d x splits %d (kchunk %d) launch %d\n", p.layer, LD[p.layer].out, LD[p.layer].in_ld, p.col0, p.col0 + p.ncols, p.shape,
             HX_WSHAPE[p.shape].bm, HX_WSHAPE[p.shape].bn, p.tiles, p.splits, p.kchunk, p.launch);
    float *slab, *bslab; CK(hipMalloc(&slab, plan.slab_floats * 4)); CK(hipMalloc(&bslab, plan.bslab_floats * 4));
    WgradMulti WM_[2];
    for (int q = 0; q < plan.nlaunch; ++q) hx_wgrad_fill(plan, q, LD, OP, ROWS, slab, bslab, WM_[q]);
    auto launch_new = [&]() { for (int q = 0; q < plan.nlaunch; ++q) hipLaunchKernelGGL(hx_wgrad_multi_kernel, dim3(hx_group_grid(WM_[q].G)), dim3(256), 0, st, WM_[q]); };
    // reference: the two production groups with the old kernel
    Case* old2[2] = {nullptr, nullptr};
    for (Case& c : cases) { if (c.name == "wgrad_critic_in") old2[0] = &c; if (c.name == "wgrad_other5") old2[1] = &c; }
    auto launch_old2 = [&]() { for (int q = 0; q < 2; ++q) old2[q]->v[0].launch(old2[q]->G, st); };
    // correctness: per layer, slices summed on the host
    CK(hipMemsetAsync(slab, 0xff, plan.slab_floats * 4, st)); CK(hipMemsetAsync(bslab, 0xff, plan.bslab_floats * 4, st));
    launch_new(); launch_old2(); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
    std::vector<float> hs(plan.slab_floats), hb(plan.bslab_floats);
    CK(hipMemcpy(hs.data(), slab, hs.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bslab, hb.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) {
      const int out = LD[i].out, in = LD[i].in_ld;
      std::vector<double> dw((size_t)out * in, 0.0), db(out, 0.0);
      for (const WgradPiece& p : plan.pieces) if (p.layer == i) {
        for (int sidx = 0; sidx < p.splits; ++sidx) for (int r = 0; r < out; ++r) for (int cc = 0; cc < p.ncols; ++cc) dw[(size_t)r * in + p.col0 + cc] += hs[p.slab_off + ((size_t)sidx * out + r) * p.ncols + cc];
        if (p.bias) for (int sidx = 0; sidx < p.splits * p.tiles_n; ++sidx) for (int r = 0; r < out; ++r) db[r] += hb[p.bslab_off + (size_t)sidx * out + r];
      }
      // the old groups' outputs: case 0 member 0 = layer order[0]; case 1 members = order[1..5]
      const Case& oc = *old2[i == 0 ? 0 : 1]; const int mi = i == 0 ? 0 : i - 1; const GemmArgs& g = oc.G.p[mi];
      std::vector<float> o((size_t)g.splits * out * in), ob((size_t)g.splits * g.db_parts * out);
      CK(hipMemcpy(o.data(), g.C, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ob.data(), g.dbias, ob.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; double maxd = 0;
      for (size_t e = 0; e < (size_t)out * in; ++e) { double r = 0; for (int sidx = 0; sidx < g.splits; ++sidx) r += o[(size_t)sidx * out * in + e]; const double d = fabs(r - dw[e]); if (!(d <= 1e-3 * (1 + fabs(r)))) ++bad; maxd = std::max(maxd, d); }
      for (int e = 0; e < out; ++e) { double r = 0; for (int sidx = 0; sidx < g.splits * g.db_parts; ++sidx) r += ob[(size_t)sidx * out + e]; const double d = fabs(r - db[e]); if (!(d <= 1e-3 * (1 + fabs(r)))) ++bad; maxd = std::max(maxd, d); }
      printf("wgrad_all layer %d (%d x %d): %zu values differ from the production kernel (max |d| %.3g)%s\n", i, out, in, bad, maxd, bad ? "   <-- MISMATCH" : "");
    }
    std::vector<float> t_old, t_new;
    for (int r = 0; r < rounds + 1; ++r) {
      float t;
      CK(hipEventRecord(e0, st)); for (int i = 0; i < per; ++i) launch_old2(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); if (r) t_old.push_back(t / per);
      CK(hipEventRecord(e0, st)); for (int i = 0; i < per; ++i) launch_new(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); if (r) t_new.push_back(t / per);
    }
    std::sort(t_old.begin(), t_old.end()); std::sort(t_new.begin(), t_new.end());
    const double fl = old2[0]->flops + old2[1]->flops;
    printf("wgrad_all        old: two grouped launches  median %8.1f us  %6.1f TF (%.3f)\n", t_old[t_old.size() / 2] * 1e3, fl / t_old[t_old.size() / 2] / 1e9, fl / t_old[t_old.size() / 2] / 1e9 / 157.3);
    printf("wgrad_all        new: %d planned launches    median %8.1f us  %6.1f TF (%.3f)\n", plan.nlaunch, t_new[t_new.size() / 2] * 1e3, fl / t_new[t_new.size() / 2] / 1e9, fl / t_new[t_new.size() / 2] / 1e9 / 157.3);
    for (int q = 0; q < plan.nlaunch; ++q) {
      std::vector<float> tq;
      for (int r = 0; r < rounds; ++r) { float t; CK(hipEventRecord(e0, st)); for (int i = 0; i < per; ++i) hipLaunchKernelGGL(hx_wgrad_multi_kernel, dim3(hx_group_grid(WM_[q].G)), dim3(256), 0, st, WM_[q]); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); tq.push_back(t / per); }
      std::sort(tq.begin(), tq.end());
      double f = 0; for (const WgradPiece& p : plan.pieces) if (p.launch == q) f += 2.0 * LD[p.layer].out * p.ncols * ROWS;
      printf("wgrad_all        new launch %d: grid %d  median %8.1f us  %6.1f TF (%.3f)\n", q, hx_group_grid(WM_[q].G), tq[tq.size() / 2] * 1e3, f / tq[tq.size() / 2] / 1e9, f / tq[tq.size() / 2] / 1e9 / 157.3);
    }
    fflush(stdout);
  }
  for (Case& c : cases) {
    if (filter[0] && c.name.find(filter) == std::string::npos) continue;
    // per-variant copy of the work description: split-K products are cut for the variant's tile shape and slot count
    const bool slab = (c.G.p[0].splits > 0);
    std::vector<GemmGroup> GV(c.v.size(), c.G);
    for (size_t vi = 0; vi < c.v.size(); ++vi) if (slab) {
      int T = 0;
      for (int i = 0; i < c.G.n; ++i) T += ((c.G.p[i].M + c.v[vi].bm - 1) / c.v[vi].bm) * ((c.G.p[i].N + c.v[vi].bn - 1) / c.v[vi].bn);
      int splits = std::min(MAXSPLITS, std::max(1, c.v[vi].slots / T));
      const int kchunk = rup((ROWS + splits - 1) / splits, 32); splits = (ROWS + kchunk - 1) / kchunk;
      for (int i = 0; i < c.G.n; ++i) { GemmArgs& g = GV[vi].p[i]; g.splits = splits; g.kchunk = kchunk; g.db_parts = (g.N + c.v[vi].bn - 1) / c.v[vi].bn; }
      printf("# %-16s %-22s %d tiles x %d slices = %d workgroups on %d slots, kchunk %d: K loop of a wave = %.0f k cycles\n", c.name.c_str(), c.v[vi].name.c_str(), T, splits, T * splits, c.v[vi].slots, kchunk,
             (c.v[vi].bm / 32.0) * (c.v[vi].bn / 32.0) / 4.0 * kchunk * 32.0 / 1e3);
    }
    // correctness: every variant against variant 0 (bit for bit; split-K products: slices summed on the host, compared to 1e-3)
    std::vector<std::vector<double>> ref;
    for (size_t vi = 0; vi < c.v.size(); ++vi) {
      for (auto& o : c.outs) CK(hipMemsetAsync(o.first, 0xff, o.second * 4, st));
      c.v[vi].launch(GV[vi], st);
      CK(hipStreamSynchronize(st)); CK(hipGetLastError());
      size_t bad = 0, total = 0; double maxd = 0;
      for (size_t oi = 0; oi < c.outs.size(); ++oi) {
        std::vector<float> h(c.outs[oi].second);
        CK(hipMemcpy(h.data(), c.outs[oi].first, h.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> red;
        if (slab) {
          const GemmArgs& g = GV[vi].p[oi / 2];
          const bool isdb = (oi & 1);
          const size_t n = isdb ? (size_t)g.M : (size_t)g.M * g.ldc;
          const int parts = isdb ? g.splits * g.db_parts : g.splits;
          red.assign(n, 0.0);
          for (int sidx = 0; sidx < parts; ++sidx) for (size_t i = 0; i < n; ++i) red[i] += h[(size_t)sidx * n + i];
        } else red.assign(h.begin(), h.end());
        if (vi == 0) ref.push_back(red);
        else {
          for (size_t i = 0; i < red.size(); ++i) {
            const double d = fabs(red[i] - ref[oi][i]);
            const bool tol = slab || c.v[vi].name.compare(0, 2, "sk") == 0;      // split tiles of a stream-K launch: another association of the same sum
            if (tol ? (d > 1e-3 * (1.0 + fabs(ref[oi][i])) || red[i] != red[i]) : (d != 0.0 || red[i] != red[i])) ++bad;
            if (d > maxd) maxd = d;
          }
          total += red.size();
        }
      }
      if (vi) printf("%-16s %-22s vs variant 0: %zu of %zu values differ (max |d| %.3g)%s\n", c.name.c_str(), c.v[vi].name.c_str(), bad, total, maxd, bad ? "   <-- MISMATCH" : "");
    }
    // timing: interleaved rounds
    std::vector<std::vector<float>> ms(c.v.size());
    for (int r = 0; r < rounds + 1; ++r)
      for (size_t vi = 0; vi < c.v.size(); ++vi) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < per; ++i) c.v[vi].launch(GV[vi], st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r) ms[vi].push_back(t / per);
      }
    for (size_t vi = 0; vi < c.v.size(); ++vi) {
      std::sort(ms[vi].begin(), ms[vi].end());
      const float med = ms[vi][ms[vi].size() / 2], mn = ms[vi][0];
      printf("%-16s %-22s median %8.1f us  %6.1f TF (%.3f)   min %8.1f us %6.1f TF\n", c.name.c_str(), c.v[vi].name.c_str(), med * 1e3, c.flops / med / 1e9, c.flops / med / 1e9 / 157.3,
             mn * 1e3, c.flops / mn / 1e9);
    }
    fflush(stdout);
  }
  if (g_sk_err) { int e = 0; CK(hipMemcpy(&e, g_sk_err, 4, hipMemcpyDeviceToHost)); printf("stream-K wait budget exhausted: %s\n", e ? "YES  <-- ERROR" : "no"); }
  return 0;
}
