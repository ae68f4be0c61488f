// hx_dyn.h -- per-lane articulated-body dynamics of the hector biped (device code, fp32).
//
// TWO lanes per environment: lane 2e owns the left leg of robot e, lane 2e+1 the right leg; both carry the
// floating base redundantly.  The hector tree is two 5-body chains hanging off the base, so Featherstone's
// articulated-body algorithm splits cleanly: each lane runs the leaf-to-root recursion of its own chain
// and the two chains' contributions to the base (6x6 articulated inertia + bias force, 33 floats) are
// summed with one lane-pair exchange (__shfl_xor 1).  Per-leg constants (joint offsets, inertias, limits,
// collision corners) are staged once per workgroup in LDS (hx_model_data.h HXM_LEGC) and read with the
// lane's leg offset, which keeps the leg code identical for both lanes -- half the instruction footprint
// of a one-lane-per-robot unrolling, which did not fit the instruction cache (DESIGN.md "Env-step kernel").
// Three linearly-implicit terms are folded into the articulated inertias (DESIGN.md "Physics model"):
//   * ground contact at the shape corner points:  f = f0 - B a_body   (B = sum Xc^T K Xc, 6x6 PSD)
//   * PD actuation (reference legged_robot.py:339-355) while unclipped:  D_i += dt (Kd + dt Kp)
//   * soft joint limits:  D_i += dt (d + dt k)
// The independent float64 statement of the same equations (joint-space CRBA + RNEA + dense solve) is
// oracle/physics.py; tests/test_sim_parity.py compares the two.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "hx_model_data.h"
#include "hx_model_data_full.h"

#define HXD __device__ __forceinline__

struct V3 { float x, y, z; };
HXD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
HXD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
HXD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
HXD V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
HXD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
HXD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HXD V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HXD float get(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct M3 { float m[9]; };   // row-major
HXD M3 m3zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.f; return r; }
HXD V3 row(const M3& a, int i) { return mk(a.m[3 * i], a.m[3 * i + 1], a.m[3 * i + 2]); }
HXD V3 col(const M3& a, int j) { return mk(a.m[j], a.m[3 + j], a.m[6 + j]); }
HXD void setrow(M3& a, int i, V3 v) { a.m[3 * i] = v.x; a.m[3 * i + 1] = v.y; a.m[3 * i + 2] = v.z; }
HXD void setcol(M3& a, int j, V3 v) { a.m[j] = v.x; a.m[3 + j] = v.y; a.m[6 + j] = v.z; }
HXD V3 mul(const M3& a, V3 v) { return mk(dot(row(a, 0), v), dot(row(a, 1), v), dot(row(a, 2), v)); }
HXD V3 mulT(const M3& a, V3 v) { return mk(dot(col(a, 0), v), dot(col(a, 1), v), dot(col(a, 2), v)); }
HXD M3 operator+(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
HXD M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * j + i]; return r; }
// r x A  (cross of r with every column of A)
HXD M3 crossM(V3 r, const M3& a) { M3 o; for (int j = 0; j < 3; ++j) setcol(o, j, cross(r, col(a, j))); return o; }
HXD void addouter(M3& a, float s, V3 u, V3 v) {
  a.m[0] += s * u.x * v.x; a.m[1] += s * u.x * v.y; a.m[2] += s * u.x * v.z;
  a.m[3] += s * u.y * v.x; a.m[4] += s * u.y * v.y; a.m[5] += s * u.y * v.z;
  a.m[6] += s * u.z * v.x; a.m[7] += s * u.z * v.y; a.m[8] += s * u.z * v.z;
}

// rotation about coordinate axis K by angle with (c,s):  R v  and  R^T v
template <int K> HXD V3 rot(float c, float s, V3 v) {
  if (K == 0) return mk(v.x, c * v.y - s * v.z, s * v.y + c * v.z);
  if (K == 1) return mk(c * v.x + s * v.z, v.y, -s * v.x + c * v.z);
  return mk(c * v.x - s * v.y, s * v.x + c * v.y, v.z);
}
template <int K> HXD V3 rotT(float c, float s, V3 v) { return rot<K>(c, -s, v); }
// R A R^T
template <int K> HXD M3 rotM(float c, float s, const M3& a) {
  M3 b;
  for (int j = 0; j < 3; ++j) setcol(b, j, rot<K>(c, s, col(a, j)));
  M3 o;
  for (int i = 0; i < 3; ++i) setrow(o, i, rot<K>(c, s, row(b, i)));
  return o;
}

// sin/cos for joint angles.  Joint ranges are within +-2.3 rad (URDF limits + soft-limit overshoot), so the
// argument is wrapped to [-pi, pi] (a no-op for any limited joint), folded into [-pi/2, pi/2] and evaluated
// with Taylor polynomials (|err| < 6e-8 there).  This replaces sincosf's generic Payne-Hanek slow path,
// which alone was ~1/6 of the kernel's instruction footprint.
HXD void joint_sincos(float x, float* s, float* c) {
  const float PI = 3.14159265358979f;
  x = fmaf(-6.28318530717959f, rintf(x * 0.159154943091895f), x);
  float sgn = 1.f;
  if (x > 0.5f * PI) { x = PI - x; sgn = -1.f; }
  else if (x < -0.5f * PI) { x = -PI - x; sgn = -1.f; }
  const float x2 = x * x;
  float ps = -2.50521084e-8f;                 // -1/11!
  ps = fmaf(ps, x2, 2.75573192e-6f);          //  1/9!
  ps = fmaf(ps, x2, -1.98412698e-4f);         // -1/7!
  ps = fmaf(ps, x2, 8.33333333e-3f);          //  1/5!
  ps = fmaf(ps, x2, -1.66666667e-1f);         // -1/3!
  *s = fmaf(ps * x2, x, x);
  float pc = 2.08767570e-9f;                  //  1/12!
  pc = fmaf(pc, x2, -2.75573192e-7f);         // -1/10!
  pc = fmaf(pc, x2, 2.48015873e-5f);          //  1/8!
  pc = fmaf(pc, x2, -1.38888889e-3f);         // -1/6!
  pc = fmaf(pc, x2, 4.16666667e-2f);          //  1/4!
  pc = fmaf(pc, x2, -0.5f);
  *c = sgn * fmaf(pc, x2, 1.0f);
}

struct SV { V3 w, v; };      // spatial motion [angular; linear] or force [moment; force]
HXD SV operator+(SV a, SV b) { SV r; r.w = a.w + b.w; r.v = a.v + b.v; return r; }
HXD SV operator-(SV a, SV b) { SV r; r.w = a.w - b.w; r.v = a.v - b.v; return r; }
struct SI { M3 A, H, M; };   // 6x6 symmetric [[A,H],[H^T,M]]
HXD SV mulSI(const SI& I, SV a) { SV f; f.w = mul(I.A, a.w) + mul(I.H, a.v); f.v = mulT(I.H, a.w) + mul(I.M, a.v); return f; }


template <typename F, int... Is> HXD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> HXD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

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
};

#define HX_LEG_NJ 5
#define HX_ARM_NJ 4
// Kinematic chains hanging off the base on one body side.  Joint frames are axis aligned (no rpy in the URDF joint origins).
//   leg: hip yaw z, hip roll x, thigh / calf / toe pitch y; collision shapes on the thigh (slot 0) and the toe (slot 1)
//   arm (hector_full only): shoulder yaw z, pitch y, roll x, elbow pitch y; shapes on twist (2), shoulder (3), elbow (4)
struct LegChain {
  static constexpr int NJ = HX_LEG_NJ, B0 = 0, NS = 2;
  static constexpr int axis(int L) { return (L == 0) ? 2 : (L == 1) ? 0 : 1; }
  static constexpr int slot(int L) { return (L == 2) ? 0 : (L == 4) ? 1 : -1; }     // collision-shape slot of local body L
  static constexpr int slot0 = 0;
};
struct ArmChain {
  static constexpr int NJ = HX_ARM_NJ, B0 = HX_LEG_NJ, NS = 3;
  static constexpr int axis(int L) { return (L == 0) ? 2 : (L == 1) ? 1 : (L == 2) ? 0 : 1; }
  static constexpr int slot(int L) { return (L == 0) ? 2 : (L == 1) ? 3 : (L == 3) ? 4 : -1; }
  static constexpr int slot0 = 2;
};
// the two robots of the family: joints per lane (= per body side) and where their constants live
struct ModelHector {
  static constexpr int NL = HX_LEG_NJ, NB = HX_LEG_NJ, NSHAPE = 2; static constexpr bool ARMS = false;
  static constexpr int STRIDE = HX_LEGC_STRIDE;
  HXD static const float* side_table() { return HXM_LEGC; }
  HXD static const float* base_pts() { return HXM_CONTACT_PTS; }
  HXD static float io(int k) { return HXM_IO[k]; }
  HXD static float h(int k) { return HXM_H[k]; }
  HXD static float mass0() { return HXM_MASS[0]; }
};
struct ModelFull {          // hector with arms (task hector_full): one leg and one arm per body side
  static constexpr int NL = HX_LEG_NJ + HX_ARM_NJ, NB = HX_LEG_NJ + HX_ARM_NJ, NSHAPE = 5; static constexpr bool ARMS = true;
  static constexpr int STRIDE = HXF_SIDE_STRIDE;
  HXD static const float* side_table() { return HXF_SIDEC; }
  HXD static const float* base_pts() { return HXF_BASE_PTS; }
  HXD static float io(int k) { return HXF_IO[k]; }
  HXD static float h(int k) { return HXF_H[k]; }
  HXD static float mass0() { return HXF_MASS0; }
};
template <int L> struct LegAxis { static constexpr int value = LegChain::axis(L); };
#define HX_LDS_CONST_FLOATS_OF(M) (2 * M::STRIDE + 24)
#define HX_LDS_CONST_FLOATS HX_LDS_CONST_FLOATS_OF(ModelHector)

// stage the per-side table and the base collision corners into LDS (call with all threads, then __syncthreads)
template <class M> HXD void dyn_stage_constants(float* lds, int tid, int nthreads) {
  for (int i = tid; i < 2 * M::STRIDE; i += nthreads) lds[i] = M::side_table()[i];
  for (int i = tid; i < 24; i += nthreads) lds[2 * M::STRIDE + i] = M::base_pts()[i];
}

// this lane's view of the constants: NB bodies of 16 floats, then the collision corner blocks (24 floats per shape slot)
template <class M> struct SideConst {
  const float* t;      // LDS, side table of this lane
  const float* basept; // LDS, 8 base corners
  HXD V3 off(int b) const { return mk(t[b * 16], t[b * 16 + 1], t[b * 16 + 2]); }
  HXD V3 h(int b) const { return mk(t[b * 16 + 3], t[b * 16 + 4], t[b * 16 + 5]); }
  HXD float mass(int b) const { return t[b * 16 + 12]; }
  HXD float qlo(int b) const { return t[b * 16 + 13]; }
  HXD float qhi(int b) const { return t[b * 16 + 14]; }
  HXD float vmax(int b) const { return t[b * 16 + 15]; }
  HXD SI inertia(int b) const {
    SI r;
    const float xx = t[b * 16 + 6], yy = t[b * 16 + 7], zz = t[b * 16 + 8], xy = t[b * 16 + 9], xz = t[b * 16 + 10], yz = t[b * 16 + 11];
    r.A.m[0] = xx; r.A.m[1] = xy; r.A.m[2] = xz; r.A.m[3] = xy; r.A.m[4] = yy; r.A.m[5] = yz; r.A.m[6] = xz; r.A.m[7] = yz; r.A.m[8] = zz;
    const V3 hh = h(b);
    r.H.m[0] = 0.f; r.H.m[1] = -hh.z; r.H.m[2] = hh.y; r.H.m[3] = hh.z; r.H.m[4] = 0.f; r.H.m[5] = -hh.x; r.H.m[6] = -hh.y; r.H.m[7] = hh.x; r.H.m[8] = 0.f;
    const float m = mass(b);
    r.M = m3zero(); r.M.m[0] = m; r.M.m[4] = m; r.M.m[8] = m;
    return r;
  }
  HXD const float* pts(int slot) const { return t + M::NB * 16 + slot * 24; }
};
typedef SideConst<ModelHector> LegConst;

// v x* (I v) for a rigid body with spatial inertia `in` (H = skew(h), M = m 1)
HXD SV rb_bias(const SI& in, V3 hh, float m, SV v) {
  const V3 hw = mul(in.A, v.w) + cross(hh, v.v);
  const V3 hv = m * v.v + cross(v.w, hh);
  SV p; p.w = cross(v.w, hw) + cross(v.v, hv); p.v = cross(v.w, hv);
  return p;
}

template <class M> HXD SI base_inertia(float s) {
  SI r;
  const float xx = M::io(0), yy = M::io(1), zz = M::io(2), xy = M::io(3), xz = M::io(4), yz = M::io(5);
  r.A.m[0] = s * xx; r.A.m[1] = s * xy; r.A.m[2] = s * xz; r.A.m[3] = s * xy; r.A.m[4] = s * yy; r.A.m[5] = s * yz;
  r.A.m[6] = s * xz; r.A.m[7] = s * yz; r.A.m[8] = s * zz;
  const V3 hh = s * mk(M::h(0), M::h(1), M::h(2));
  r.H.m[0] = 0.f; r.H.m[1] = -hh.z; r.H.m[2] = hh.y; r.H.m[3] = hh.z; r.H.m[4] = 0.f; r.H.m[5] = -hh.x; r.H.m[6] = -hh.y; r.H.m[7] = hh.x; r.H.m[8] = 0.f;
  const float m = s * M::mass0();
  r.M = m3zero(); r.M.m[0] = m; r.M.m[4] = m; r.M.m[8] = m;
  return r;
}

// Height and unit normal of the terrain triangle under world (x, y): every grid cell is split along its
// (i,j)-(i+1,j+1) diagonal (the split of convert_heightfield_to_trimesh); oracle/terrain.py HeightField.query.
HXD float terrain_query(const DynParams& P, float u, float w, V3& nw) {
  const int i = min(max((int)floorf(u), 0), HX_PATCH - 2), j = min(max((int)floorf(w), 0), HX_PATCH - 2);
  const float fu = fminf(fmaxf(u - (float)i, 0.f), 1.f), fw = fminf(fmaxf(w - (float)j, 0.f), 1.f);
  const float* c = P.patch + i * HX_PATCH + j;
  const float h00 = c[0], h01 = c[1], h10 = c[HX_PATCH], h11 = c[HX_PATCH + 1];
  const bool upper = fw > fu;
  const float gu = upper ? h11 - h01 : h10 - h00;
  const float gw = upper ? h01 - h00 : h11 - h10;
  const float nx = -gu * P.inv_hs, ny = -gw * P.inv_hs;
  const float inv = 1.0f / sqrtf(nx * nx + ny * ny + 1.0f);
  nw = mk(nx * inv, ny * inv, inv);
  return h00 + fu * gu + fw * gw;
}

// Accumulate the contact terms of `npts` corner points (LDS, xyz triples) on a body with spatial velocity v
// (body coords), body->world rotation Rb and world position pb of the body origin.
// a_true == nullptr: f0 += explicit spatial force, B += implicit 6x6.
// a_true != nullptr: returns the implicit-consistent net force (body coords)  sum_c [f0_c - K_c Xc a].
// Plane: normal = world z, penetration = -z.  Terrain: normal of the triangle under the point, penetration =
// distance to that triangle's plane.
HXD V3 contact_points(const DynParams& P, const float* pts, int npts, SV v, const M3& Rb, V3 pb, SV& f0, SI& B, const SV* a_true) {
  const float c_n = P.dn + P.kn * P.dt;
  V3 net = mk(0.f, 0.f, 0.f);
  const V3 zb = row(Rb, 2);          // world z in body coords
#pragma unroll 1
  for (int k = 0; k < npts; ++k) {
    const V3 r = mk(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
    const float z = pb.z + dot(zb, r);
    V3 nb = zb;
    float pen = -z;
    if (P.patch != nullptr) {
      if (!__any(z < P.zmax)) continue;
      // patch coordinates of the point; over the central part of the window the tighter bound applies
      const float u = (pb.x + dot(row(Rb, 0), r) - P.px0) * P.inv_hs, w = (pb.y + dot(row(Rb, 1), r) - P.py0) * P.inv_hs;
      const float lo = (float)(HX_PATCH / 4), hi = (float)(HX_PATCH - HX_PATCH / 4);
      const bool near = (u >= lo) && (u <= hi) && (w >= lo) && (w <= hi);
      if (!__any(z < (near ? P.zmax_near : P.zmax))) continue;
      V3 nw;
      const float h = terrain_query(P, u, w, nw);
      pen = (h - z) * nw.z;
      nb = mulT(Rb, nw);
    }
    const V3 vp = v.v + cross(v.w, r);
    const float vn = dot(vp, nb);
    const float fn0 = P.kn * pen - c_n * vn;
    const bool act = (pen > 0.f) && (fn0 > 0.f);
    if (!__any(act)) continue;
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

HXD float xchg(float x) { return __shfl_xor(x, 1); }     // the other leg's lane of the same robot
HXD V3 xchg(V3 a) { return mk(xchg(a.x), xchg(a.y), xchg(a.z)); }

template <class M> struct DynStateT {
  V3 pos; float quat[4];   // base: xyzw, body->world (identical on both lanes of a robot)
  V3 linvel, angvel;       // base, world frame
  float q[M::NL], qd[M::NL];   // this lane's side: leg joints, then (hector_full) arm joints
};
typedef DynStateT<ModelHector> DynState;

HXD M3 quat_to_mat(const float* q) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  M3 r;
  r.m[0] = 1.f - 2.f * (y * y + z * z); r.m[1] = 2.f * (x * y - z * w); r.m[2] = 2.f * (x * z + y * w);
  r.m[3] = 2.f * (x * y + z * w); r.m[4] = 1.f - 2.f * (x * x + z * z); r.m[5] = 2.f * (y * z - x * w);
  r.m[6] = 2.f * (x * z - y * w); r.m[7] = 2.f * (y * z + x * w); r.m[8] = 1.f - 2.f * (x * x + y * y);
  return r;
}

// world-frame net contact forces: base = whole-base total (same on both lanes), shape[slot] = this lane's shapes
// (slot 0 thigh, 1 toe, and with arms 2 twist, 3 shoulder, 4 elbow)
template <class M> struct SideForcesT { V3 base; V3 shape[M::NSHAPE]; };
typedef SideForcesT<ModelHector> LegForces;

// per-chain working set of one substep
template <class CH> struct ChainWork {
  SV v[CH::NJ + 1];                            // [0] = base, [L+1] = local body L
  float cs_c[CH::NJ + 1], cs_s[CH::NJ + 1];
  SV U[CH::NJ + 1]; float Dinv[CH::NJ + 1], uu[CH::NJ + 1];
  M3 Rs[CH::NS]; V3 ps[CH::NS];                // body -> world rotation / origin of the bodies that carry collision shapes
  SV a[CH::NJ + 1];                            // accelerations relative to the gravity field
};

// ---- pass 1: kinematics down the chain
template <class CH, class M>
HXD void chain_pass1(ChainWork<CH>& W, const DynStateT<M>& S, const SideConst<M>& C, SV v0, const M3& R0) {
  W.v[0] = v0;
  M3 Rc = R0; V3 pc = S.pos;
  static_for<CH::NJ>([&](auto ic) {
    constexpr int L = decltype(ic)::value;      // local body 0.., state index L+1
    constexpr int K = CH::axis(L);
    constexpr int B = CH::B0 + L;               // body / joint index on this side
    float s, c;
    joint_sincos(S.q[B], &s, &c);
    W.cs_c[L + 1] = c; W.cs_s[L + 1] = s;
    const V3 r = C.off(B);
    const V3 t = W.v[L].v + cross(W.v[L].w, r);
    W.v[L + 1].w = rotT<K>(c, s, W.v[L].w);
    W.v[L + 1].v = rotT<K>(c, s, t);
    if (K == 0) W.v[L + 1].w.x += S.qd[B];
    if (K == 1) W.v[L + 1].w.y += S.qd[B];
    if (K == 2) W.v[L + 1].w.z += S.qd[B];
    pc = pc + mul(Rc, r);
    for (int i = 0; i < 3; ++i) setrow(Rc, i, rotT<K>(c, s, row(Rc, i)));
    if constexpr (CH::slot(L) >= 0) { W.Rs[CH::slot(L) - CH::slot0] = Rc; W.ps[CH::slot(L) - CH::slot0] = pc; }
  });
}

// ---- pass 2: articulated inertias, leaf -> root of the chain; hands (accI, accP) to the base.
// target/kp/kd/tau_lim/tau_out are indexed by the side-local joint index B.
template <class CH, class M>
HXD void chain_pass2(ChainWork<CH>& W, const DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, const float* target,
                     const float* kp, const float* kd, const float* tau_lim, float* tau_out, SI& accI, SV& accP) {
  accI.A = m3zero(); accI.H = m3zero(); accI.M = m3zero();
  accP.w = mk(0, 0, 0); accP.v = mk(0, 0, 0);
  static_for<CH::NJ>([&](auto ic) {
    constexpr int L = CH::NJ - 1 - decltype(ic)::value;     // last .. 0
    constexpr int K = CH::axis(L);
    constexpr int B = CH::B0 + L;
    SI IA = C.inertia(B);
    SV pA = rb_bias(IA, C.h(B), C.mass(B), W.v[L + 1]);
    if (L < CH::NJ - 1) {
      IA.A = IA.A + accI.A; IA.H = IA.H + accI.H; IA.M = IA.M + accI.M;
      pA = pA + accP;
    }
    if constexpr (CH::slot(L) >= 0) {
      constexpr int SL = CH::slot(L);
      const M3& Rb = W.Rs[SL - CH::slot0];
      SV f0; f0.w = mk(0, 0, 0); f0.v = mk(0, 0, 0);
      SI Bc; Bc.A = m3zero(); Bc.H = m3zero(); Bc.M = m3zero();
      contact_points(P, C.pts(SL), 8, W.v[L + 1], Rb, W.ps[SL - CH::slot0], f0, Bc, nullptr);
      IA.A = IA.A + Bc.A; IA.H = IA.H + Bc.H; IA.M = IA.M + Bc.M;
      SV g; g.w = mk(0, 0, 0); g.v = P.gz * row(Rb, 2);
      pA = pA - f0 + mulSI(Bc, g);
    }
    // joint-space terms: PD torque (reference legged_robot.py:339-355) + soft limits, linearly implicit
    const float q = S.q[B], qd = S.qd[B];
    const float raw = kp[B] * (target[B] - q) - kd[B] * qd;
    const float tau = fminf(fmaxf(raw, -tau_lim[B]), tau_lim[B]);
    tau_out[B] = tau;
    float beta = (raw == tau) ? P.dt * (kd[B] + P.dt * kp[B]) : 0.f;
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
    W.U[L + 1] = Ui; W.Dinv[L + 1] = di; W.uu[L + 1] = ui;
    // Ia = IA - U U^T / D ; pa = pA + Ia c + U ui / D
    addouter(IA.A, -di, Ui.w, Ui.w);
    addouter(IA.H, -di, Ui.w, Ui.v);
    addouter(IA.M, -di, Ui.v, Ui.v);
    SV cI;   // c_i = v_i x (S qd)
    {
      const V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
      cI.w = cross(W.v[L + 1].w, w2); cI.v = cross(W.v[L + 1].v, w2);
    }
    SV pa = pA + mulSI(IA, cI);
    pa.w = pa.w + (ui * di) * Ui.w; pa.v = pa.v + (ui * di) * Ui.v;
    // transform to the parent frame:  X^T Ia X,  X^T pa
    const float c = W.cs_c[L + 1], s = W.cs_s[L + 1];
    const V3 r = C.off(B);
    const M3 A1 = rotM<K>(c, s, IA.A), H1 = rotM<K>(c, s, IA.H), M1 = rotM<K>(c, s, IA.M);
    const M3 G = crossM(r, M1);                       // rx M'
    const M3 T1 = crossM(r, transpose(H1));           // rx H'^T
    const M3 Kk = crossM(r, transpose(G));            // rx G^T = (G rx^T)^T, symmetric
    accI.A = A1 + T1 + transpose(T1) + Kk;
    accI.H = H1 + G;
    accI.M = M1;
    accP.v = rot<K>(c, s, pa.v);
    accP.w = rot<K>(c, s, pa.w) + cross(r, accP.v);
  });
}

// ---- pass 3: accelerations, root -> leaf (W.a[0] = base acceleration)
template <class CH, class M>
HXD void chain_pass3(ChainWork<CH>& W, const DynStateT<M>& S, const SideConst<M>& C, float* qdd) {
  static_for<CH::NJ>([&](auto ic) {
    constexpr int L = decltype(ic)::value;
    constexpr int K = CH::axis(L);
    constexpr int B = CH::B0 + L;
    const float c = W.cs_c[L + 1], s = W.cs_s[L + 1];
    const V3 r = C.off(B);
    const float qd = S.qd[B];
    SV ai;
    ai.w = rotT<K>(c, s, W.a[L].w);
    ai.v = rotT<K>(c, s, W.a[L].v + cross(W.a[L].w, r));
    const V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
    ai.w = ai.w + cross(W.v[L + 1].w, w2);
    ai.v = ai.v + cross(W.v[L + 1].v, w2);
    const float dd = W.Dinv[L + 1] * (W.uu[L + 1] - (dot(W.U[L + 1].w, ai.w) + dot(W.U[L + 1].v, ai.v)));
    qdd[B] = dd;
    if (K == 0) ai.w.x += dd;
    if (K == 1) ai.w.y += dd;
    if (K == 2) ai.w.z += dd;
    W.a[L + 1] = ai;
  });
}

// implicit-consistent net contact forces of the chain's shapes (world frame), last substep of an env step only
template <class CH, class M>
HXD void chain_forces(const ChainWork<CH>& W, const DynParams& P, const SideConst<M>& C, V3* shape_force) {
  static_for<CH::NJ>([&](auto ic) {
    constexpr int L = decltype(ic)::value;
    if constexpr (CH::slot(L) >= 0) {
      constexpr int SL = CH::slot(L);
      const M3& Rb = W.Rs[SL - CH::slot0];
      SV dmy; SI dmyB;
      SV at = W.a[L + 1]; at.v = at.v + P.gz * row(Rb, 2);     // true spatial acceleration
      shape_force[SL] = mul(Rb, contact_points(P, C.pts(SL), 8, W.v[L + 1], Rb, W.ps[SL - CH::slot0], dmy, dmyB, &at));
    }
  });
}

// One 1 ms substep of this lane's side of the robot.  target/kp/kd/tau_lim/tau_out: this side's joints (leg, then arm).
template <class M>
HXD void dyn_substep(DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, int leg, const float* target, const float* kp,
                     const float* kd, const float* tau_lim, float mass_scale, float* tau_out, bool want_forces, SideForcesT<M>& F) {
  ChainWork<LegChain> WL;
  ChainWork<ArmChain> WA;                      // untouched (and removed by the compiler) without arms
  const M3 R0 = quat_to_mat(S.quat);
  const V3 nb_base = row(R0, 2);
  SV v0; v0.w = mulT(R0, S.angvel); v0.v = mulT(R0, S.linvel);
  chain_pass1<LegChain, M>(WL, S, C, v0, R0);
  if constexpr (M::ARMS) chain_pass1<ArmChain, M>(WA, S, C, v0, R0);
  SI accI; SV accP;
  chain_pass2<LegChain, M>(WL, S, P, C, target, kp, kd, tau_lim, tau_out, accI, accP);
  if constexpr (M::ARMS) {
    SI aI; SV aP;
    chain_pass2<ArmChain, M>(WA, S, P, C, target, kp, kd, tau_lim, tau_out, aI, aP);
    accI.A = accI.A + aI.A; accI.H = accI.H + aI.H; accI.M = accI.M + aI.M;
    accP = accP + aP;
  }
  // ---- base: each lane adds HALF of the base corners to its side's contribution, then the two lanes of the
  //      robot exchange and sum (a + b == b + a bitwise, so both lanes hold the identical base system)
  const SV g0 = [&] { SV g; g.w = mk(0, 0, 0); g.v = P.gz * nb_base; return g; }();
  {
    SV f0; f0.w = mk(0, 0, 0); f0.v = mk(0, 0, 0);
    SI B; B.A = m3zero(); B.H = m3zero(); B.M = m3zero();
    contact_points(P, C.basept + 12 * leg, 4, v0, R0, S.pos, f0, B, nullptr);
    accI.A = accI.A + B.A; accI.H = accI.H + B.H; accI.M = accI.M + B.M;
    accP = accP - f0 + mulSI(B, g0);
  }
  SI baseI = base_inertia<M>(mass_scale);
  SV baseP = rb_bias(baseI, mass_scale * mk(M::h(0), M::h(1), M::h(2)), mass_scale * M::mass0(), v0);
  for (int i = 0; i < 9; ++i) {
    baseI.A.m[i] += accI.A.m[i] + xchg(accI.A.m[i]);
    baseI.H.m[i] += accI.H.m[i] + xchg(accI.H.m[i]);
    baseI.M.m[i] += accI.M.m[i] + xchg(accI.M.m[i]);
  }
  baseP.w = baseP.w + (accP.w + xchg(accP.w));
  baseP.v = baseP.v + (accP.v + xchg(accP.v));
  SV a0;
  {
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
    a0.w = mk(bm[0], bm[1], bm[2]);
    a0.v = mk(bm[3], bm[4], bm[5]);
  }
  float qdd[M::NL];
  WL.a[0] = a0;
  chain_pass3<LegChain, M>(WL, S, C, qdd);
  if constexpr (M::ARMS) { WA.a[0] = a0; chain_pass3<ArmChain, M>(WA, S, C, qdd); }
  // ---- net contact forces (implicit-consistent), last substep of an env step only
  if (want_forces) {
    SV dmy; SI dmyB;
    {
      SV at = a0; at.v = at.v + g0.v;
      const V3 part = contact_points(P, C.basept + 12 * leg, 4, v0, R0, S.pos, dmy, dmyB, &at);
      F.base = mul(R0, part + xchg(part));
    }
    chain_forces<LegChain, M>(WL, P, C, F.shape);
    if constexpr (M::ARMS) chain_forces<ArmChain, M>(WA, P, C, F.shape);
  }
  // ---- integrate (semi-implicit Euler); the base update is identical on both lanes
  {
    const V3 a_ang = a0.w;
    const V3 a_lin = a0.v + g0.v + cross(v0.w, v0.v);
    S.angvel = S.angvel + P.dt * mul(R0, a_ang);
    S.linvel = S.linvel + P.dt * mul(R0, a_lin);
    for (int j = 0; j < M::NL; ++j) {
      const float nqd = S.qd[j] + P.dt * qdd[j];
      const float vm = C.vmax(j);
      S.qd[j] = fminf(fmaxf(nqd, -vm), vm);
      S.q[j] += P.dt * S.qd[j];
    }
    S.pos = S.pos + P.dt * S.linvel;
    const V3 w = S.angvel;
    const float x = S.quat[0], y = S.quat[1], z = S.quat[2], ww = S.quat[3];
    const float h = 0.5f * P.dt;
    float nx = x + h * (w.x * ww + w.y * z - w.z * y);
    float ny = y + h * (w.y * ww + w.z * x - w.x * z);
    float nz = z + h * (w.z * ww + w.x * y - w.y * x);
    float nw = ww - h * (w.x * x + w.y * y + w.z * z);
    const float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
    S.quat[0] = nx * inv; S.quat[1] = ny * inv; S.quat[2] = nz * inv; S.quat[3] = nw * inv;
  }
}

// Forward kinematics for the observation/reward glue: world position / velocity of the body origins of this
// leg's calf ("knee") and toe ("foot") -- reference hector_config.py:31-32.
struct BodyOut { V3 pos, linvel, angvel; float quat[4]; };
HXD void mat_to_quat(const M3& R, float* q) {
  const float m00 = R.m[0], m11 = R.m[4], m22 = R.m[8];
  const float c0 = 1 + m00 - m11 - m22, c1 = 1 - m00 + m11 - m22, c2 = 1 - m00 - m11 + m22, c3 = 1 + m00 + m11 + m22;
  float qx, qy, qz, qw;
  if (c3 >= c0 && c3 >= c1 && c3 >= c2) {
    const float t4 = 2.f * sqrtf(fmaxf(c3, 1e-30f));
    qw = 0.25f * t4; qx = (R.m[7] - R.m[5]) / t4; qy = (R.m[2] - R.m[6]) / t4; qz = (R.m[3] - R.m[1]) / t4;
  } else if (c0 >= c1 && c0 >= c2) {
    const float t4 = 2.f * sqrtf(fmaxf(c0, 1e-30f));
    qx = 0.25f * t4; qy = (R.m[1] + R.m[3]) / t4; qz = (R.m[2] + R.m[6]) / t4; qw = (R.m[7] - R.m[5]) / t4;
  } else if (c1 >= c2) {
    const float t4 = 2.f * sqrtf(fmaxf(c1, 1e-30f));
    qx = (R.m[1] + R.m[3]) / t4; qy = 0.25f * t4; qz = (R.m[5] + R.m[7]) / t4; qw = (R.m[2] - R.m[6]) / t4;
  } else {
    const float t4 = 2.f * sqrtf(fmaxf(c2, 1e-30f));
    qx = (R.m[2] + R.m[6]) / t4; qy = (R.m[5] + R.m[7]) / t4; qz = 0.25f * t4; qw = (R.m[3] - R.m[1]) / t4;
  }
  const float sg = qw < 0.f ? -1.f : 1.f;
  q[0] = sg * qx; q[1] = sg * qy; q[2] = sg * qz; q[3] = sg * qw;
}
template <class M> HXD void dyn_body_states(const DynStateT<M>& S, const SideConst<M>& C, BodyOut& calf, BodyOut& toe) {
  M3 Rc = quat_to_mat(S.quat);
  V3 pc = S.pos;
  SV vc; vc.w = mulT(Rc, S.angvel); vc.v = mulT(Rc, S.linvel);
  static_for<HX_LEG_NJ>([&](auto ic) {
    constexpr int L = decltype(ic)::value;
    constexpr int K = LegAxis<L>::value;
    float s, c;
    joint_sincos(S.q[L], &s, &c);
    const V3 r = C.off(L);
    const V3 t = vc.v + cross(vc.w, r);
    vc.w = rotT<K>(c, s, vc.w);
    vc.v = rotT<K>(c, s, t);
    if (K == 0) vc.w.x += S.qd[L];
    if (K == 1) vc.w.y += S.qd[L];
    if (K == 2) vc.w.z += S.qd[L];
    pc = pc + mul(Rc, r);
    for (int i = 0; i < 3; ++i) setrow(Rc, i, rotT<K>(c, s, row(Rc, i)));
    if (L == 3 || L == 4) {
      BodyOut& o = (L == 3) ? calf : toe;
      o.pos = pc;
      o.linvel = mul(Rc, vc.v);
      o.angvel = mul(Rc, vc.w);
      mat_to_quat(Rc, o.quat);
    }
  });
}
