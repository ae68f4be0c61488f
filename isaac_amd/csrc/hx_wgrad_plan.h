// hx_wgrad_plan.h -- host side of hx_wgrad_multi_kernel (hx_gemm_sp.h): cuts the weight-gradient products of a minibatch
// (dW_l[out][in] = dZ_l^T X_l, reduction over the minibatch rows; rollout_storage / ppo.py:150-172 of the reference through autograd)
// into launches of at most one workgroup per CU whose workgroups all take the same time.
//
// A layer becomes one or two PIECES (column ranges of dW): a main piece in the menu shape that pads it least and, when the
// layer's width is not a whole number of tiles, a remainder piece (a narrow strip or a 128-wide tile).  Every piece gets its own
// split count, slab region [splits][out x ncols] (compact) and reduce segment; the pieces of all layers are assigned to one or
// two launches by exhaustive search over the assignments (<= 2^8), split counts by a greedy that always gives the next slice to
// the piece whose workgroups are longest while the launch still fits the CUs.  Runs once per learner (hx_ppo_create).
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>
#include "hx_gemm_sp.h"

struct WgradLayerDesc { int out, in_ld; };
struct WgradPiece {
  int layer, col0, ncols, shape, tiles, splits, kchunk, launch;
  bool bias;                 // this piece also produces the layer's bias gradient (column sums of dZ): the main piece
  size_t slab_off, bslab_off;   // float offsets of [splits][out * ncols] and [splits * tiles_n][out]
  int tiles_n;
};
struct WgradPlan { std::vector<WgradPiece> pieces; int nlaunch = 0; size_t slab_floats = 0, bslab_floats = 0; double est_unit_rows = 0; };

#ifndef HX_WP_EPI_ROWS
#define HX_WP_EPI_ROWS 150
#endif
static inline int hx_wp_rup(int a, int b) { return (a + b - 1) / b * b; }
// cycles per MFMA of a shape's K loop relative to shape 0's (profiles/r04_b_gemm_lab.txt, "cal_" cases: measured launch time of one
// product cut for 256 workgroups over the nominal 64 cycles per MFMA, epilogue taken out): smaller wave tiles issue more
// fragment reads and staging instructions per MFMA and have a shorter load-to-use distance
static inline double hx_wp_penalty(int shape) { return shape == 4 ? 1.30 : shape == 3 ? 1.16 : shape == 2 ? 1.10 : shape == 1 ? 1.07 : 1.0; }
static inline int hx_wp_tiles(int shape, int out, int ncols) {
  return ((out + HX_WSHAPE[shape].bm - 1) / HX_WSHAPE[shape].bm) * ((ncols + HX_WSHAPE[shape].bn - 1) / HX_WSHAPE[shape].bn);
}
// padded work of a piece in (32x32 tiles) x penalty
static inline double hx_wp_work(int shape, int out, int ncols) { return hx_wp_tiles(shape, out, ncols) * 4.0 * HX_WSHAPE[shape].units * hx_wp_penalty(shape); }

// split counts of the pieces of one launch (indices `idx` into P); returns the makespan in unit-rows (units x rows per split x penalty)
static inline double hx_wp_balance(std::vector<WgradPiece>& P, const std::vector<int>& idx, int Mrows, int slots, int max_splits) {
  if (idx.empty()) return 0.0;
  int used = 0;
  for (int i : idx) { P[i].splits = 1; used += P[i].tiles; }
  if (used > slots) return 1e30;
  std::vector<char> frozen(P.size(), 0);
  // a workgroup's time in unit-rows: its K loop plus what it costs to start and to write 16 accumulator registers per unit
  // to its slab, expressed as rows of the K loop (HX_WP_EPI_ROWS, from profiles/r04_b_gemm_lab.txt: launches of 120- and 240-tile loops)
  auto cost = [&](int i) { return HX_WSHAPE[P[i].shape].units * hx_wp_penalty(P[i].shape) * (double)(hx_wp_rup((Mrows + P[i].splits - 1) / P[i].splits, 32) + HX_WP_EPI_ROWS); };
  for (;;) {
    int j = -1; double cj = -1;
    for (int i : idx) { const double c = cost(i); if (c > cj) { cj = c; j = i; } }
    if (frozen[j]) break;
    const int ns = P[j].splits + 1;
    if (used + P[j].tiles <= slots && ns <= max_splits && (Mrows + ns - 1) / ns >= 256) { P[j].splits = ns; used += P[j].tiles; }
    else frozen[j] = 1;
  }
  double mk = 0;
  for (int i : idx) {
    P[i].kchunk = hx_wp_rup((Mrows + P[i].splits - 1) / P[i].splits, 32);
    P[i].splits = (Mrows + P[i].kchunk - 1) / P[i].kchunk;
    mk = std::max(mk, cost(i));
  }
  return mk;
}

static inline WgradPlan hx_wgrad_plan(const WgradLayerDesc* L, int nl, int Mrows, int slots, int max_splits = 32) {
  WgradPlan plan;
  std::vector<WgradPiece>& P = plan.pieces;
  for (int l = 0; l < nl; ++l) {
    const int out = L[l].out, in = L[l].in_ld;
    // best single shape for the whole layer
    int best = 3; double bw = 1e30;
    for (int sh = 0; sh < 4; ++sh) { const double w = hx_wp_work(sh, out, in); if (w < bw) { bw = w; best = sh; } }
    // or: whole tiles of a main shape + a remainder piece
    int ms = -1, rs = -1, mcols = 0; double sw = bw;
    for (int sh = 0; sh < 4; ++sh) {
      const int bn = HX_WSHAPE[sh].bn, mc = in / bn * bn;
      if (mc == 0 || mc == in) continue;
      for (int r = 1; r < HX_WSHAPES; ++r) {
        if (r == 4 && in - mc > 32) continue;
        const double w = hx_wp_work(sh, out, mc) + hx_wp_work(r, out, in - mc);
        if (w < sw * 0.97) { sw = w; ms = sh; rs = r; mcols = mc; }      // a remainder piece must buy at least 3 %
      }
    }
    if (ms < 0) P.push_back(WgradPiece{l, 0, in, best, hx_wp_tiles(best, out, in), 1, 0, 0, true, 0, 0, (in + HX_WSHAPE[best].bn - 1) / HX_WSHAPE[best].bn});
    else {
      P.push_back(WgradPiece{l, 0, mcols, ms, hx_wp_tiles(ms, out, mcols), 1, 0, 0, true, 0, 0, mcols / HX_WSHAPE[ms].bn});
      P.push_back(WgradPiece{l, mcols, in - mcols, rs, hx_wp_tiles(rs, out, in - mcols), 1, 0, 0, false, 0, 0, (in - mcols + HX_WSHAPE[rs].bn - 1) / HX_WSHAPE[rs].bn});
    }
  }
  const int np = (int)P.size();
  // assignment of the pieces to one or two launches (piece 0 always in launch 0)
  double best_t = 1e30; unsigned best_mask = 0;
  const double launch_cost = 1500.0;        // unit-rows: ~20 us of launch / drain / slab write per launch
  for (unsigned mask = 0; mask < (1u << (np - 1)); ++mask) {
    std::vector<int> a, b;
    for (int i = 0; i < np; ++i) (((mask << 1) >> i) & 1 ? b : a).push_back(i);
    if ((int)a.size() > HX_GROUP_MAX || (int)b.size() > HX_GROUP_MAX) continue;
    std::vector<WgradPiece> Q = P;
    const double t = hx_wp_balance(Q, a, Mrows, slots, max_splits) + hx_wp_balance(Q, b, Mrows, slots, max_splits) + launch_cost * (b.empty() ? 1 : 2);
    if (t < best_t) { best_t = t; best_mask = mask; }
  }
  std::vector<int> a, b;
  for (int i = 0; i < np; ++i) { const bool second = ((best_mask << 1) >> i) & 1; P[i].launch = second; (second ? b : a).push_back(i); }
  hx_wp_balance(P, a, Mrows, slots, max_splits);
  hx_wp_balance(P, b, Mrows, slots, max_splits);
  plan.nlaunch = b.empty() ? 1 : 2;
  plan.est_unit_rows = best_t;
  for (WgradPiece& p : P) {
    p.slab_off = plan.slab_floats; plan.slab_floats += (size_t)p.splits * L[p.layer].out * p.ncols;
    p.bslab_off = plan.bslab_floats; if (p.bias) plan.bslab_floats += (size_t)p.splits * p.tiles_n * L[p.layer].out;
  }
  return plan;
}

// kernel arguments of launch `launch` of a plan: operands of layer l are dZ[l] ([Mrows][out]) and X[l] ([Mrows][ldx[l]])
struct WgradOperands { const float* dZ; const float* X; int ldx; };
static inline void hx_wgrad_fill(const WgradPlan& plan, int launch, const WgradLayerDesc* L, const WgradOperands* op, int Mrows, float* slab, float* bslab,
                                 WgradMulti& W) {
  W = WgradMulti{};
  int n = 0, blocks = 0;
  for (const WgradPiece& p : plan.pieces) {
    if (p.launch != launch) continue;
    GemmArgs& g = W.G.p[n];
    const int out = L[p.layer].out;
    g.A = op[p.layer].dZ; g.lda = out; g.B = op[p.layer].X + p.col0; g.ldb = op[p.layer].ldx;
    g.C = slab + p.slab_off; g.ldc = p.ncols; g.M = out; g.N = p.ncols; g.K = Mrows;
    g.splits = p.splits; g.kchunk = p.kchunk;
    g.dbias = p.bias ? bslab + p.bslab_off : nullptr; g.db_parts = p.tiles_n;
    g.tiles_m = (out + HX_WSHAPE[p.shape].bm - 1) / HX_WSHAPE[p.shape].bm; g.tiles_n = p.tiles_n;
    W.shape[n] = p.shape;
    W.G.first[n] = blocks; blocks += p.tiles * p.splits;
    ++n;
  }
  W.G.first[n] = blocks; W.G.n = n;
}
