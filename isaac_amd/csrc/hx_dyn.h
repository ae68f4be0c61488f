// hx_dyn.h -- articulated-body dynamics of the hector family of bipeds (fp32), single source for the gfx950 kernels
// and the host build (hx_math.h explains the split).
//
// Device layout: TWO lanes per robot.  Lane 2e owns the left body side of robot e, lane 2e+1 the right side; both carry
// the floating base redundantly.  Every robot of the family is a base with kinematic chains hanging off it (hector: one
// 5-joint leg per side; hector_full: a leg and a 4-joint arm per side; XBot-L: one 6-joint leg per side), so
// Featherstone's articulated-body algorithm splits cleanly per side: a lane runs the leaf-to-root recursion of its own
// chains and the two sides' contributions to the base (6x6 articulated inertia + bias force, 33 floats) are summed with
// one lane-pair exchange (DPP quad permutation, hx_xchg).  Per-side constants (joint offsets, inertias, limits, PD gains,
// collision points) are staged once per workgroup in LDS and read with the lane's side offset, which keeps the code
// identical for both lanes -- half the instruction footprint of a one-lane-per-robot unrolling, which did not fit the
// instruction cache (DESIGN.md "Env-step kernel").  Nothing inside the substep loop reads global memory.
//
// Three linearly-implicit terms are folded into the articulated inertias (DESIGN.md "Physics model"):
//   * ground contact at the collision points:  f = f0 - B a_body   (B = sum Xc^T K Xc, 6x6 PSD)
//   * PD actuation (reference legged_robot.py:339-355) while unclipped:  D_i += dt (Kd + dt Kp)
//   * soft joint limits:  D_i += dt (d + dt k)
// The independent float64 statement of the same equations (joint-space CRBA + RNEA + dense solve) is
// oracle/physics.py; tests/test_host_build.py (CPU) and tests/test_gpu_sim.py compare against it.
#pragma once
#include "hx_math.h"
#include "hx_model_data.h"
#include "hx_model_data_full.h"
#include "hx_model_data_xbot.h"

// ---- the robots of the family: generated tables (tools/compile_urdf.py) + what the task glue needs to know about them
struct ModelHector : HXM_Hector {
  static constexpr bool ARMS = false;
  static constexpr int KNEE = 3, FOOT = 4;       // side-local bodies named by asset.knee_name / foot_name (hector_config.py:31-32)
  HXD static const float* side_table() { return HXM_SIDE; }
  HXD static const float* base_table() { return HXM_BASE; }
  static constexpr float MASS0 = HXM_MASS0;
};
struct ModelFull : HXM_Full {             // hector with arms (task hector_full): leg bodies 0-4, arm bodies 5-8 per side
  static constexpr int KNEE = 3, FOOT = 4;
  HXD static const float* side_table() { return HXF_SIDE; }
  HXD static const float* base_table() { return HXF_BASE; }
  static constexpr float MASS0 = HXF_MASS0;
};
struct ModelXBot : HXM_XBot {             // XBot-L (task humanoid_ppo): roll, yaw, pitch, knee, ankle pitch, ankle roll
  static constexpr int KNEE = 3, FOOT = 5;       // 'knee' / 'ankle_roll' (humanoid_config.py:64-65)
  HXD static const float* side_table() { return HXX_SIDE; }
  HXD static const float* base_table() { return HXX_BASE; }
  static constexpr float MASS0 = HXX_MASS0;
};
template <class M> struct ModelInfo {
  static constexpr int nshape() { int n = 0; for (int k = 0; k < M::NL; ++k) n += M::NPTS[k] > 0 ? 1 : 0; return n; }
  static constexpr int slot(int B) { int n = 0; for (int k = 0; k < B; ++k) n += M::NPTS[k] > 0 ? 1 : 0; return M::NPTS[B] > 0 ? n : -1; }
  static constexpr int chain_start_of(int B) { int s = 0; for (int c = 0; c < M::NCH; ++c) if (B >= M::CH_START[c] && B < M::CH_START[c] + M::CH_LEN[c]) s = M::CH_START[c]; return s; }
  static constexpr int NSHAPE = nshape();
  static constexpr int LDS_FLOATS = 2 * M::SIDE_STRIDE + M::BASE_FLOATS + 2 * M::NL * 4;
};

// Terrain: each robot keeps a HX_PATCH x HX_PATCH window of the height grid (metres, fp32) in LDS, centred on its
// base at the start of the env step; `patch == nullptr` selects the ground plane z = 0.
#define HX_PATCH 16
struct DynParams {
  float dt, gz, kn, dn, veps, lim_k, lim_d, mu;
  const float* patch;     // LDS, [HX_PATCH][HX_PATCH] row-major (row = x index), or nullptr
  float px0, py0;         // world x / y of patch node (0, 0)
  float inv_hs;           // 1 / horizontal_scale
  float zmax;             // highest node of the patch: points above it cannot touch
  float zmax_near;        // highest node of the central (HX_PATCH/2 + 1)^2 nodes: the bound for points over that part
  float wall;             // height difference between grid neighbours beyond which the trimesh has a vertical wall
                          // (slope_treshold * horizontal_scale, reference utils/terrain.py:70-73); 0 = no walls (heightfield)
};

// stage the per-side tables, the base table and the PD constants of both sides (pd_src: kp[ND], kd[ND], tau_lim[ND],
// default_pos[ND] in DoF order = left side then right side) into LDS; call with all threads, then synchronise
template <class M> HXD void dyn_stage_constants(float* lds, int tid, int nthreads, const float* kp, const float* kd, const float* tl, const float* q0) {
  for (int i = tid; i < 2 * M::SIDE_STRIDE; i += nthreads) lds[i] = M::side_table()[i];
  for (int i = tid; i < M::BASE_FLOATS; i += nthreads) lds[2 * M::SIDE_STRIDE + i] = M::base_table()[i];
  float* pd = lds + 2 * M::SIDE_STRIDE + M::BASE_FLOATS;
  for (int i = tid; i < 2 * M::NL; i += nthreads) { pd[4 * i] = kp[i]; pd[4 * i + 1] = kd[i]; pd[4 * i + 2] = tl[i]; pd[4 * i + 3] = q0[i]; }
}

// one side's view of the constants
template <class M> struct SideConst {
  const float* t;      // side table
  const float* bt;     // base table: sphere 4, points 3 * NBASE, inertia 6, h 3, mass
  const float* pd;     // [NL][4] kp, kd, tau_lim, default position of this side's joints
  HXD void bind(const float* lds, int side) { t = lds + side * M::SIDE_STRIDE; bt = lds + 2 * M::SIDE_STRIDE; pd = bt + M::BASE_FLOATS + side * M::NL * 4; }
  HXD V3 off(int b) const { return ld3(t + b * M::JSTRIDE); }
  HXD V3 h(int b) const { return ld3(t + b * M::JSTRIDE + 3); }
  HXD float mass(int b) const { return t[b * M::JSTRIDE + 12]; }
  HXD float qlo(int b) const { return t[b * M::JSTRIDE + 13]; }
  HXD float qhi(int b) const { return t[b * M::JSTRIDE + 14]; }
  HXD float vmax(int b) const { return t[b * M::JSTRIDE + 15]; }
  HXD M3 rotc(int b) const { return ld9(t + b * M::JSTRIDE + 16); }      // child -> parent rotation at q = 0 (HAS_ROT models)
  HXD float kp(int b) const { return pd[4 * b]; }
  HXD float kd(int b) const { return pd[4 * b + 1]; }
  HXD float tau_lim(int b) const { return pd[4 * b + 2]; }
  HXD float q0(int b) const { return pd[4 * b + 3]; }
  HXD static SI spatial(const float* io, V3 hh, float m, float s) {
    SI r;
    r.A.m[0] = s * io[0]; r.A.m[1] = s * io[3]; r.A.m[2] = s * io[4]; r.A.m[3] = s * io[3]; r.A.m[4] = s * io[1]; r.A.m[5] = s * io[5];
    r.A.m[6] = s * io[4]; r.A.m[7] = s * io[5]; r.A.m[8] = s * io[2];
    hh = s * hh;
    r.H.m[0] = 0.f; r.H.m[1] = -hh.z; r.H.m[2] = hh.y; r.H.m[3] = hh.z; r.H.m[4] = 0.f; r.H.m[5] = -hh.x; r.H.m[6] = -hh.y; r.H.m[7] = hh.x; r.H.m[8] = 0.f;
    r.M = m3zero(); r.M.m[0] = s * m; r.M.m[4] = s * m; r.M.m[8] = s * m;
    return r;
  }
  HXD SI inertia(int b) const { return spatial(t + b * M::JSTRIDE + 6, h(b), mass(b), 1.0f); }
  HXD const float* shape(int b) const { return t + M::PTS_OFF[b]; }        // sphere (centre xyz, radius), then the points
  HXD const float* base_shape() const { return bt; }
  HXD V3 base_h() const { return ld3(bt + 4 + 3 * M::NBASE + 6); }
  HXD float base_mass() const { return bt[4 + 3 * M::NBASE + 9]; }
  HXD SI base_inertia(float s) const { return spatial(bt + 4 + 3 * M::NBASE, base_h(), base_mass(), s); }
};

// v x* (I v) for a rigid body with spatial inertia `in` (H = skew(h), M = m 1)
HXD SV rb_bias(const SI& in, V3 hh, float m, SV v) {
  const V3 hw = mul(in.A, v.w) + cross(hh, v.v);
  const V3 hv = m * v.v + cross(v.w, hh);
  SV p; p.w = cross(v.w, hw) + cross(v.v, hv); p.v = cross(v.w, hv);
  return p;
}

// Height and unit normal of the terrain surface under patch coordinates (u, w).  Every grid cell is split along its
// (i,j)-(i+1,j+1) diagonal (the split of convert_heightfield_to_trimesh); oracle/terrain.py HeightField.query.
// With P.wall > 0 (mesh_type 'trimesh') a cell that contains a height jump larger than P.wall is a cell whose low
// vertices the reference moved under the high ones (slope_treshold, utils/terrain.py:70-73): the low ground continues
// flat through the cell and a vertical wall stands on the high vertices' grid line.  Such a cell returns its low level
// here; the wall itself is handled by wall_push() for points that have crossed it.
HXD float terrain_query(const DynParams& P, float u, float w, V3& nw) {
  const int i = hx_imin(hx_imax((int)floorf(u), 0), HX_PATCH - 2), j = hx_imin(hx_imax((int)floorf(w), 0), HX_PATCH - 2);
  const float fu = fminf(fmaxf(u - (float)i, 0.f), 1.f), fw = fminf(fmaxf(w - (float)j, 0.f), 1.f);
  const float* c = P.patch + i * HX_PATCH + j;
  float h00 = c[0], h01 = c[1], h10 = c[HX_PATCH], h11 = c[HX_PATCH + 1];
  if (P.wall > 0.f) {
    const float lo = fminf(fminf(h00, h01), fminf(h10, h11));
    if (fmaxf(fmaxf(h00, h01), fmaxf(h10, h11)) - lo > P.wall) {
      // vertices standing more than a wall height above the cell's lowest one are "high": the mesh has no surface of
      // theirs inside this cell, their place is taken by the low level
      h00 = (h00 - lo > P.wall) ? lo : h00; h01 = (h01 - lo > P.wall) ? lo : h01;
      h10 = (h10 - lo > P.wall) ? lo : h10; h11 = (h11 - lo > P.wall) ? lo : h11;
    }
  }
  const bool upper = fw > fu;
  const float gu = upper ? h11 - h01 : h10 - h00;
  const float gw = upper ? h01 - h00 : h11 - h10;
  const float nx = -gu * P.inv_hs, ny = -gw * P.inv_hs;
  const float inv = 1.0f / sqrtf(nx * nx + ny * ny + 1.0f);
  nw = mk(nx * inv, ny * inv, inv);
  return h00 + fu * gu + fw * gw;
}

// A point below the surface of a plateau may have entered it sideways through a wall.  If one of the four sides of its
// cell is the top line of a wall (the neighbouring cell in that direction lies more than P.wall lower at the nearer grid
// line) and the point is closer to that wall than to the surface above it, the contact is with the wall: returns true
// and replaces (pen, nw) by the horizontal distance and the wall's outward normal.
HXD bool wall_push(const DynParams& P, float u, float w, float z, float& pen, V3& nw) {
  const int i = hx_imin(hx_imax((int)floorf(u), 1), HX_PATCH - 3), j = hx_imin(hx_imax((int)floorf(w), 1), HX_PATCH - 3);
  const float fu = fminf(fmaxf(u - (float)i, 0.f), 1.f), fw = fminf(fmaxf(w - (float)j, 0.f), 1.f);
  const int jn = j + (fw > 0.5f ? 1 : 0), in_ = i + (fu > 0.5f ? 1 : 0);        // the nearer grid line across the step
  const float* g = P.patch;
  const float hs = 1.0f / P.inv_hs;
  float best = pen; int dir = -1;
  // -u side: wall on grid line i if node (i-1, jn) lies a wall height below node (i, jn) and the point is below that top
  { const float top = g[i * HX_PATCH + jn], low = g[(i - 1) * HX_PATCH + jn]; const float d = fu * hs;
    if (top - low > P.wall && z < top && z > low - 0.5f * P.wall && d < best) { best = d; dir = 0; } }
  { const float top = g[(i + 1) * HX_PATCH + jn], low = g[(i + 2) * HX_PATCH + jn]; const float d = (1.f - fu) * hs;
    if (top - low > P.wall && z < top && z > low - 0.5f * P.wall && d < best) { best = d; dir = 1; } }
  { const float top = g[in_ * HX_PATCH + j], low = g[in_ * HX_PATCH + j - 1]; const float d = fw * hs;
    if (top - low > P.wall && z < top && z > low - 0.5f * P.wall && d < best) { best = d; dir = 2; } }
  { const float top = g[in_ * HX_PATCH + j + 1], low = g[in_ * HX_PATCH + j + 2]; const float d = (1.f - fw) * hs;
    if (top - low > P.wall && z < top && z > low - 0.5f * P.wall && d < best) { best = d; dir = 3; } }
  if (dir < 0) return false;
  pen = best;
  nw = (dir == 0) ? mk(-1.f, 0.f, 0.f) : (dir == 1) ? mk(1.f, 0.f, 0.f) : (dir == 2) ? mk(0.f, -1.f, 0.f) : mk(0.f, 1.f, 0.f);
  return true;
}

// Accumulate the contact terms of points [p0, p0 + npts) of a shape (`shp`: bounding sphere, then xyz triples; LDS) on a
// body with spatial velocity v (body coords), body->world rotation Rb and world position pb of the body origin.
// a_true == nullptr: f0 += explicit spatial force, B += implicit 6x6.
// a_true != nullptr: returns the implicit-consistent net force (body coords)  sum_c [f0_c - K_c Xc a].
// Plane: normal = world z, penetration = -z.  Terrain: normal of the triangle under the point, penetration = distance to
// that triangle's plane; or the wall the point went through (wall_push).
HXD V3 contact_points(const DynParams& P, const float* shp, int p0, int npts, SV v, const M3& Rb, V3 pb, SV& f0, SI& B, const SV* a_true) {
  V3 net = mk(0.f, 0.f, 0.f);
  const V3 zb = row(Rb, 2);          // world z in body coords
  {
    // bounding sphere of the whole shape: nothing of it can touch while its lowest possible point is above the ground's
    // highest one (0 on the plane, the patch maximum on terrain)
    const float zc = pb.z + dot(zb, ld3(shp)) - shp[3];
    if (!hx_any(zc < ((P.patch != nullptr) ? P.zmax : 0.f))) return net;
  }
  const float c_n = P.dn + P.kn * P.dt;
  const float* pts = shp + 4;
#pragma unroll 1
  for (int k = p0; k < p0 + npts; ++k) {
    const V3 r = ld3(pts + 3 * k);
    const float z = pb.z + dot(zb, r);
    V3 nb = zb;
    float pen = -z;
    if (P.patch != nullptr) {
      if (!hx_any(z < P.zmax)) continue;
      // patch coordinates of the point; over the central part of the window the tighter bound applies
      const float u = (pb.x + dot(row(Rb, 0), r) - P.px0) * P.inv_hs, w = (pb.y + dot(row(Rb, 1), r) - P.py0) * P.inv_hs;
      const float lo = (float)(HX_PATCH / 4), hi = (float)(HX_PATCH - HX_PATCH / 4);
      const bool near = (u >= lo) && (u <= hi) && (w >= lo) && (w <= hi);
      if (!hx_any(z < (near ? P.zmax_near : P.zmax))) continue;
      V3 nw;
      const float h = terrain_query(P, u, w, nw);
      pen = (h - z) * nw.z;
      if (P.wall > 0.f && hx_any(pen > 0.f)) {
        float wp = pen; V3 wn = nw;
        if (pen > 0.f && wall_push(P, u, w, z, wp, wn)) { pen = wp; nw = wn; }
      }
      nb = mulT(Rb, nw);
    }
    const V3 vp = v.v + cross(v.w, r);
    const float vn = dot(vp, nb);
    const float fn0 = P.kn * pen - c_n * vn;
    const bool act = (pen > 0.f) && (fn0 > 0.f);
    if (!hx_any(act)) continue;
    const V3 vt = vp - vn * nb;
    const float vtn = sqrtf(dot(vt, vt));
    const float c_t = P.mu * fn0 / fmaxf(vtn, P.veps);
    const float on = act ? 1.f : 0.f;
    const V3 f = on * (fn0 * nb - c_t * vt);
    const float alpha = on * P.dt * c_t, beta = on * P.dt * (c_n - c_t);
    if (a_true == nullptr) {
      f0.v = f0.v + f;
      f0.w = f0.w + cross(r, f);
      const V3 m = cross(r, nb);
      const float rr = dot(r, r);
      // A += alpha (|r|^2 1 - r r^T) + beta m m^T ; H += alpha rx + beta m n^T ; M += alpha 1 + beta n n^T
      B.A.m[0] += alpha * rr; B.A.m[4] += alpha * rr; B.A.m[8] += alpha * rr;
      addouter(B.A, -alpha, r, r);
      addouter(B.A, beta, m, m);
      B.H.m[1] += -alpha * r.z; B.H.m[2] += alpha * r.y;
      B.H.m[3] += alpha * r.z;  B.H.m[5] += -alpha * r.x;
      B.H.m[6] += -alpha * r.y; B.H.m[7] += alpha * r.x;
      addouter(B.H, beta, m, nb);
      B.M.m[0] += alpha; B.M.m[4] += alpha; B.M.m[8] += alpha;
      addouter(B.M, beta, nb, nb);
    } else {
      const V3 ap = a_true->v + cross(a_true->w, r);          // Xc a
      const V3 ka = alpha * ap + (beta * dot(nb, ap)) * nb;    // K Xc a
      net = net + (f - ka);
    }
  }
  return net;
}

// 6x6 SPD solve (Cholesky, fully unrolled, static indices): x = A^-1 b
HXD void solve6(float (&a)[6][6], float (&b)[6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int k = 0; k < j; ++k) a[j][j] -= a[j][k] * a[j][k];
    const float d = sqrtf(a[j][j]);
    const float inv = 1.0f / d;
    a[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
#pragma unroll
      for (int k = 0; k < j; ++k) a[i][j] -= a[i][k] * a[j][k];
      a[i][j] *= inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) b[i] -= a[i][k] * b[k];
    b[i] /= a[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < 6; ++k) b[i] -= a[k][i] * b[k];
    b[i] /= a[i][i];
  }
}

// state of one body side: the floating base (identical on both sides of a robot) and the side's joints
template <class M> struct DynStateT {
  V3 pos; float quat[4];   // base: xyzw, body->world
  V3 linvel, angvel;       // base, world frame
  float q[M::NL], qd[M::NL];
};

// world-frame net contact forces: base = this side's HALF of the base points (the driver sums the two halves),
// shape[slot] = the side's bodies that carry collision points, in body order
template <class M> struct SideForcesT { V3 base; V3 shape[ModelInfo<M>::NSHAPE]; };

// per-side working set of one substep
template <class M> struct SideWork {
  static constexpr int NL = M::NL, NS = ModelInfo<M>::NSHAPE;
  M3 R0; SV v0, g0;                             // base rotation, base velocity (base coords), gravity field in base coords
  SV v[NL];                                      // body velocities, body coords
  float cs_c[NL], cs_s[NL];
  SV U[NL]; float Dinv[NL], uu[NL];
  M3 Rs[NS]; V3 ps[NS];                          // body -> world rotation / origin of the bodies that carry collision points
  SV a[NL];                                      // accelerations relative to the gravity field
  float tau[NL];
};

// ---- upward half of a substep for one side: kinematics, articulated inertias leaf -> root, this side's half of the base
// points; hands (accI, accP) = the side's contribution to the base system.  target = PD position target per joint.
template <class M>
HXD void side_up(SideWork<M>& W, const DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, int side, const float* target, SI& accI, SV& accP) {
  using MI = ModelInfo<M>;
  W.R0 = quat_to_mat(S.quat);
  W.v0.w = mulT(W.R0, S.angvel); W.v0.v = mulT(W.R0, S.linvel);
  W.g0.w = mk(0.f, 0.f, 0.f); W.g0.v = P.gz * row(W.R0, 2);
  accI = si0(); accP = sv0();
  static_for<M::NCH>([&](auto cc) {
    constexpr int CH = decltype(cc)::value, S0 = M::CH_START[CH], LEN = M::CH_LEN[CH];
    // ---- pass 1: kinematics down the chain
    {
      M3 Rc = W.R0; V3 pc = S.pos;
      static_for<LEN>([&](auto ic) {
        constexpr int B = S0 + decltype(ic)::value;
        constexpr int K = M::AXIS[B];
        float s, c;
        joint_sincos(S.q[B], &s, &c);
        W.cs_c[B] = c; W.cs_s[B] = s;
        const V3 r = C.off(B);
        const SV vp = (B == S0) ? W.v0 : W.v[B == S0 ? B : B - 1];
        V3 w = vp.w, t = vp.v + cross(vp.w, r);
        pc = pc + mul(Rc, r);
        if constexpr (M::HAS_ROT) { const M3 E = C.rotc(B); w = mulT(E, w); t = mulT(E, t); Rc = matmul(Rc, E); }
        W.v[B].w = rotT<K>(c, s, w);
        W.v[B].v = rotT<K>(c, s, t);
        if (K == 0) W.v[B].w.x += S.qd[B];
        if (K == 1) W.v[B].w.y += S.qd[B];
        if (K == 2) W.v[B].w.z += S.qd[B];
        for (int i = 0; i < 3; ++i) setrow(Rc, i, rotT<K>(c, s, row(Rc, i)));
        if constexpr (MI::slot(B) >= 0) { W.Rs[MI::slot(B)] = Rc; W.ps[MI::slot(B)] = pc; }
      });
    }
    // ---- pass 2: articulated inertias, leaf -> root of the chain
    SI chI = si0(); SV chP = sv0();
    static_for<LEN>([&](auto ic) {
      constexpr int B = S0 + LEN - 1 - decltype(ic)::value;     // last .. first
      constexpr int K = M::AXIS[B];
      SI IA = C.inertia(B);
      SV pA = rb_bias(IA, C.h(B), C.mass(B), W.v[B]);
      if (B < S0 + LEN - 1) { siadd(IA, chI); pA = pA + chP; }
      if constexpr (MI::slot(B) >= 0) {
        constexpr int SL = MI::slot(B);
        const M3& Rb = W.Rs[SL];
        SV f0 = sv0(); SI Bc = si0();
        contact_points(P, C.shape(B), 0, M::NPTS[B], W.v[B], Rb, W.ps[SL], f0, Bc, nullptr);
        siadd(IA, Bc);
        SV g; g.w = mk(0.f, 0.f, 0.f); g.v = P.gz * row(Rb, 2);
        pA = pA - f0 + mulSI(Bc, g);
      }
      // joint-space terms: PD torque (reference legged_robot.py:339-355) + soft limits, linearly implicit
      const float q = S.q[B], qd = S.qd[B];
      const float kp = C.kp(B), kd = C.kd(B), tl = C.tau_lim(B);
      const float raw = kp * (target[B] - q) - kd * qd;
      const float tau = fminf(fmaxf(raw, -tl), tl);
      W.tau[B] = tau;
      float beta = (raw == tau) ? P.dt * (kd + P.dt * kp) : 0.f;
      const float c_lim = P.lim_d + P.lim_k * P.dt;
      const float lo_pen = C.qlo(B) - q, hi_pen = q - C.qhi(B);
      const float t_lo = P.lim_k * lo_pen - c_lim * qd;
      const float t_hi = -P.lim_k * hi_pen - c_lim * qd;
      const bool act_lo = (lo_pen > 0.f) && (t_lo > 0.f);
      const bool act_hi = (hi_pen > 0.f) && (t_hi < 0.f);
      const float tau_j = tau + (act_lo ? t_lo : 0.f) + (act_hi ? t_hi : 0.f);
      beta += (act_lo || act_hi) ? c_lim * P.dt : 0.f;
      // U = IA S ; D = S^T U
      SV Ui; Ui.w = col(IA.A, K); Ui.v = row(IA.H, K);
      const float D = get(Ui.w, K) + beta;
      const float di = 1.0f / D;
      const float ui = tau_j - get(pA.w, K);
      W.U[B] = Ui; W.Dinv[B] = di; W.uu[B] = ui;
      // Ia = IA - U U^T / D ; pa = pA + Ia c + U ui / D
      addouter(IA.A, -di, Ui.w, Ui.w);
      addouter(IA.H, -di, Ui.w, Ui.v);
      addouter(IA.M, -di, Ui.v, Ui.v);
      SV cI;   // c_i = v_i x (S qd)
      {
        const V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
        cI.w = cross(W.v[B].w, w2); cI.v = cross(W.v[B].v, w2);
      }
      SV pa = pA + mulSI(IA, cI);
      pa.w = pa.w + (ui * di) * Ui.w; pa.v = pa.v + (ui * di) * Ui.v;
      // transform to the parent frame:  X^T Ia X,  X^T pa
      const float c = W.cs_c[B], s = W.cs_s[B];
      const V3 r = C.off(B);
      M3 A1 = rotM<K>(c, s, IA.A), H1 = rotM<K>(c, s, IA.H), M1 = rotM<K>(c, s, IA.M);
      V3 pv = rot<K>(c, s, pa.v), pw = rot<K>(c, s, pa.w);
      if constexpr (M::HAS_ROT) {
        const M3 E = C.rotc(B);
        A1 = simM(E, A1); H1 = simM(E, H1); M1 = simM(E, M1);
        pv = mul(E, pv); pw = mul(E, pw);
      }
      const M3 G = crossM(r, M1);                       // rx M'
      const M3 T1 = crossM(r, transpose(H1));           // rx H'^T
      const M3 Kk = crossM(r, transpose(G));            // rx G^T = (G rx^T)^T, symmetric
      chI.A = A1 + T1 + transpose(T1) + Kk;
      chI.H = H1 + G;
      chI.M = M1;
      chP.v = pv;
      chP.w = pw + cross(r, pv);
    });
    siadd(accI, chI); accP = accP + chP;
  });
  // ---- this side's half of the base points
  {
    SV f0 = sv0(); SI Bc = si0();
    contact_points(P, C.base_shape(), side * (M::NBASE / 2), M::NBASE / 2, W.v0, W.R0, S.pos, f0, Bc, nullptr);
    siadd(accI, Bc);
    accP = accP - f0 + mulSI(Bc, W.g0);
  }
}

// base system: (I_base + sum of both sides) a0 = -(p_base + sum of both sides)
template <class M>
HXD SV base_solve(const SideConst<M>& C, const SV& v0, float mass_scale, const SI& sumI, const SV& sumP) {
  SI baseI = C.base_inertia(mass_scale);
  SV baseP = rb_bias(baseI, mass_scale * C.base_h(), mass_scale * C.base_mass(), v0);
  siadd(baseI, sumI);
  baseP = baseP + sumP;
  float Am[6][6], bm[6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Am[i][j] = baseI.A.m[3 * i + j];
      Am[i][j + 3] = baseI.H.m[3 * i + j];
      Am[i + 3][j] = baseI.H.m[3 * j + i];
      Am[i + 3][j + 3] = baseI.M.m[3 * i + j];
    }
  bm[0] = -baseP.w.x; bm[1] = -baseP.w.y; bm[2] = -baseP.w.z;
  bm[3] = -baseP.v.x; bm[4] = -baseP.v.y; bm[5] = -baseP.v.z;
  solve6(Am, bm);
  SV a0; a0.w = mk(bm[0], bm[1], bm[2]); a0.v = mk(bm[3], bm[4], bm[5]);
  return a0;
}

// ---- downward half for one side: accelerations root -> leaf, (last substep) implicit-consistent contact forces,
// integration of the side's joints.  F.base receives this side's half of the base points.
template <class M>
HXD void side_down(SideWork<M>& W, DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, int side, SV a0, bool want_forces, SideForcesT<M>& F) {
  using MI = ModelInfo<M>;
  float qdd[M::NL];
  static_for<M::NCH>([&](auto cc) {
    constexpr int CH = decltype(cc)::value, S0 = M::CH_START[CH], LEN = M::CH_LEN[CH];
    static_for<LEN>([&](auto ic) {
      constexpr int B = S0 + decltype(ic)::value;
      constexpr int K = M::AXIS[B];
      const float c = W.cs_c[B], s = W.cs_s[B];
      const V3 r = C.off(B);
      const float qd = S.qd[B];
      const SV ap = (B == S0) ? a0 : W.a[B == S0 ? B : B - 1];
      V3 aw = ap.w, av = ap.v + cross(ap.w, r);
      if constexpr (M::HAS_ROT) { const M3 E = C.rotc(B); aw = mulT(E, aw); av = mulT(E, av); }
      SV ai;
      ai.w = rotT<K>(c, s, aw);
      ai.v = rotT<K>(c, s, av);
      const V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
      ai.w = ai.w + cross(W.v[B].w, w2);
      ai.v = ai.v + cross(W.v[B].v, w2);
      const float dd = W.Dinv[B] * (W.uu[B] - (dot(W.U[B].w, ai.w) + dot(W.U[B].v, ai.v)));
      qdd[B] = dd;
      if (K == 0) ai.w.x += dd;
      if (K == 1) ai.w.y += dd;
      if (K == 2) ai.w.z += dd;
      W.a[B] = ai;
    });
  });
  if (want_forces) {
    SV dmy; SI dmyB;
    {
      SV at = a0; at.v = at.v + W.g0.v;          // true spatial acceleration of the base
      F.base = mul(W.R0, contact_points(P, C.base_shape(), side * (M::NBASE / 2), M::NBASE / 2, W.v0, W.R0, S.pos, dmy, dmyB, &at));
    }
    static_for<M::NL>([&](auto ic) {
      constexpr int B = decltype(ic)::value;
      if constexpr (MI::slot(B) >= 0) {
        constexpr int SL = MI::slot(B);
        const M3& Rb = W.Rs[SL];
        SV at = W.a[B]; at.v = at.v + P.gz * row(Rb, 2);
        F.shape[SL] = mul(Rb, contact_points(P, C.shape(B), 0, M::NPTS[B], W.v[B], Rb, W.ps[SL], dmy, dmyB, &at));
      }
    });
  }
  for (int j = 0; j < M::NL; ++j) {
    const float nqd = S.qd[j] + P.dt * qdd[j];
    const float vm = C.vmax(j);
    S.qd[j] = fminf(fmaxf(nqd, -vm), vm);
    S.q[j] += P.dt * S.qd[j];
  }
}

// semi-implicit Euler step of the floating base (identical on both sides of a robot)
template <class M>
HXD void base_integrate(DynStateT<M>& S, const DynParams& P, const M3& R0, const SV& v0, const SV& g0, const SV& a0) {
  const V3 a_ang = a0.w;
  const V3 a_lin = a0.v + g0.v + cross(v0.w, v0.v);
  S.angvel = S.angvel + P.dt * mul(R0, a_ang);
  S.linvel = S.linvel + P.dt * mul(R0, a_lin);
  S.pos = S.pos + P.dt * S.linvel;
  const V3 w = S.angvel;
  const float x = S.quat[0], y = S.quat[1], z = S.quat[2], ww = S.quat[3];
  const float h = 0.5f * P.dt;
  const float nx = x + h * (w.x * ww + w.y * z - w.z * y);
  const float ny = y + h * (w.y * ww + w.z * x - w.x * z);
  const float nz = z + h * (w.z * ww + w.x * y - w.y * x);
  const float nw = ww - h * (w.x * x + w.y * y + w.z * z);
  const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
  S.quat[0] = nx * inv; S.quat[1] = ny * inv; S.quat[2] = nz * inv; S.quat[3] = nw * inv;
}

#if defined(__HIPCC__)
__device__ __forceinline__ V3 hx_xchg(V3 a) { return mk(hx_xchg(a.x), hx_xchg(a.y), hx_xchg(a.z)); }
// One 1 ms substep on the device: this lane's side, the partner lane's contribution through the DPP exchange
// (a + b == b + a bitwise, so both lanes hold the identical base system and solve it redundantly).
template <class M>
__device__ __forceinline__ void dyn_substep(DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, int side, const float* target, float mass_scale,
                     float* tau_out, bool want_forces, SideForcesT<M>& F) {
  SideWork<M> W;
  SI accI; SV accP;
  side_up<M>(W, S, P, C, side, target, accI, accP);
  for (int i = 0; i < 9; ++i) {
    accI.A.m[i] += hx_xchg(accI.A.m[i]);
    accI.H.m[i] += hx_xchg(accI.H.m[i]);
    accI.M.m[i] += hx_xchg(accI.M.m[i]);
  }
  accP.w = accP.w + hx_xchg(accP.w);
  accP.v = accP.v + hx_xchg(accP.v);
  const SV a0 = base_solve<M>(C, W.v0, mass_scale, accI, accP);
  side_down<M>(W, S, P, C, side, a0, want_forces, F);
  if (want_forces) F.base = F.base + hx_xchg(F.base);
  for (int j = 0; j < M::NL; ++j) tau_out[j] = W.tau[j];
  base_integrate<M>(S, P, W.R0, W.v0, W.g0, a0);
}
#endif

// One 1 ms substep of a whole robot, both sides in sequence (host build; SL / SR share the base fields, kept equal)
template <class M>
HXD void dyn_substep_pair(DynStateT<M>& SL, DynStateT<M>& SR, const DynParams& P, const SideConst<M>& CL, const SideConst<M>& CR,
                          const float* targetL, const float* targetR, float mass_scale, float* tauL, float* tauR, bool want_forces,
                          SideForcesT<M>& FL, SideForcesT<M>& FR) {
  SideWork<M> WL, WR;
  SI aI, bI; SV aP, bP;
  side_up<M>(WL, SL, P, CL, 0, targetL, aI, aP);
  side_up<M>(WR, SR, P, CR, 1, targetR, bI, bP);
  siadd(aI, bI); aP = aP + bP;
  const SV a0 = base_solve<M>(CL, WL.v0, mass_scale, aI, aP);
  side_down<M>(WL, SL, P, CL, 0, a0, want_forces, FL);
  side_down<M>(WR, SR, P, CR, 1, a0, want_forces, FR);
  if (want_forces) { const V3 b = FL.base + FR.base; FL.base = b; FR.base = b; }
  for (int j = 0; j < M::NL; ++j) { tauL[j] = WL.tau[j]; tauR[j] = WR.tau[j]; }
  base_integrate<M>(SL, P, WL.R0, WL.v0, WL.g0, a0);
  SR.pos = SL.pos; SR.linvel = SL.linvel; SR.angvel = SL.angvel;
  for (int k = 0; k < 4; ++k) SR.quat[k] = SL.quat[k];
}

// Forward kinematics for the observation/reward glue: world pose / velocity of the body origins of this side's knee
// and foot bodies (asset.knee_name / foot_name) -- what the reference reads from rigid_body_state.
struct BodyOut { V3 pos, linvel, angvel; float quat[4]; };
template <class M> HXD void dyn_body_states(const DynStateT<M>& S, const SideConst<M>& C, BodyOut& knee, BodyOut& foot) {
  M3 Rc = quat_to_mat(S.quat);
  V3 pc = S.pos;
  SV vc; vc.w = mulT(Rc, S.angvel); vc.v = mulT(Rc, S.linvel);
  static_for<M::CH_LEN[0]>([&](auto ic) {          // the leg is chain 0 of every model
    constexpr int L = decltype(ic)::value;
    constexpr int K = M::AXIS[L];
    float s, c;
    joint_sincos(S.q[L], &s, &c);
    const V3 r = C.off(L);
    V3 w = vc.w, t = vc.v + cross(vc.w, r);
    pc = pc + mul(Rc, r);
    if constexpr (M::HAS_ROT) { const M3 E = C.rotc(L); w = mulT(E, w); t = mulT(E, t); Rc = matmul(Rc, E); }
    vc.w = rotT<K>(c, s, w);
    vc.v = rotT<K>(c, s, t);
    if (K == 0) vc.w.x += S.qd[L];
    if (K == 1) vc.w.y += S.qd[L];
    if (K == 2) vc.w.z += S.qd[L];
    for (int i = 0; i < 3; ++i) setrow(Rc, i, rotT<K>(c, s, row(Rc, i)));
    if (L == M::KNEE || L == M::FOOT) {
      BodyOut& o = (L == M::KNEE) ? knee : foot;
      o.pos = pc;
      o.linvel = mul(Rc, vc.v);
      o.angvel = mul(Rc, vc.w);
      mat_to_quat(Rc, o.quat);
    }
  });
}
