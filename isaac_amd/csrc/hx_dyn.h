// hx_dyn.h -- per-lane articulated-body dynamics of the hector biped (device code, fp32).
//
// One environment per lane.  Featherstone's articulated-body algorithm on the 11-body tree compiled
// from the reference's URDF (hx_model_data.h), with three linearly-implicit terms folded into the
// articulated inertias (DESIGN.md "Physics model"):
//   * ground contact at the shape corner points:  f = f0 - B a_body   (B = sum Xc^T K Xc, 6x6 PSD)
//   * PD actuation (reference legged_robot.py:339-355) while unclipped:  D_i += dt (Kd + dt Kp)
//   * soft joint limits:  D_i += dt (d + dt k)
// The independent float64 statement of the same equations (joint-space CRBA + RNEA + dense solve) is
// oracle/physics.py; tests/test_sim_parity.py compares the two.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "hx_model_data.h"

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

struct SV { V3 w, v; };      // spatial motion [angular; linear] or force [moment; force]
HXD SV operator+(SV a, SV b) { SV r; r.w = a.w + b.w; r.v = a.v + b.v; return r; }
HXD SV operator-(SV a, SV b) { SV r; r.w = a.w - b.w; r.v = a.v - b.v; return r; }
struct SI { M3 A, H, M; };   // 6x6 symmetric [[A,H],[H^T,M]]
HXD SV mulSI(const SI& I, SV a) { SV f; f.w = mul(I.A, a.w) + mul(I.H, a.v); f.v = mulT(I.H, a.w) + mul(I.M, a.v); return f; }

template <int I> struct BodyC {
  static constexpr int parent = HXM_PARENT[I];
  static constexpr int axis = HXM_AXIS[I];
  HXD static V3 off() { return mk(HXM_OFFSET[3 * I], HXM_OFFSET[3 * I + 1], HXM_OFFSET[3 * I + 2]); }
  HXD static V3 h() { return mk(HXM_H[3 * I], HXM_H[3 * I + 1], HXM_H[3 * I + 2]); }
  HXD static SI inertia(float s) {   // s: per-env scale (base payload randomisation), 1 elsewhere
    SI r;
    const float xx = HXM_IO[6 * I], yy = HXM_IO[6 * I + 1], zz = HXM_IO[6 * I + 2];
    const float xy = HXM_IO[6 * I + 3], xz = HXM_IO[6 * I + 4], yz = HXM_IO[6 * I + 5];
    r.A.m[0] = s * xx; r.A.m[1] = s * xy; r.A.m[2] = s * xz;
    r.A.m[3] = s * xy; r.A.m[4] = s * yy; r.A.m[5] = s * yz;
    r.A.m[6] = s * xz; r.A.m[7] = s * yz; r.A.m[8] = s * zz;
    const V3 hh = s * h();           // H = skew(m c)
    r.H.m[0] = 0.f; r.H.m[1] = -hh.z; r.H.m[2] = hh.y;
    r.H.m[3] = hh.z; r.H.m[4] = 0.f; r.H.m[5] = -hh.x;
    r.H.m[6] = -hh.y; r.H.m[7] = hh.x; r.H.m[8] = 0.f;
    const float m = s * HXM_MASS[I];
    r.M = m3zero(); r.M.m[0] = m; r.M.m[4] = m; r.M.m[8] = m;
    return r;
  }
  // v x* (I v) for the rigid body's own inertia
  HXD static SV bias(SV v, float s) {
    const V3 hh = s * h();
    const float m = s * HXM_MASS[I];
    SI in = inertia(s);
    V3 hw = mul(in.A, v.w) + cross(hh, v.v);
    V3 hv = m * v.v + cross(v.w, hh);
    SV p; p.w = cross(v.w, hw) + cross(v.v, hv); p.v = cross(v.w, hv);
    return p;
  }
};

template <typename F, int... Is> HXD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> HXD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct DynParams {
  float dt, gz, kn, dn, veps, lim_k, lim_d, mu;
};

// contact shapes: index into HXM_CONTACT_BODY ; shape s has 8 corner points
template <int I> struct ShapeOf { static constexpr int value = (I == 0) ? 0 : (I == 3) ? 1 : (I == 5) ? 2 : (I == 8) ? 3 : (I == 10) ? 4 : -1; };

// Accumulate the contact terms of shape SH on a body with spatial velocity v (body coords), world
// z-axis in body coords nb, world height of the body origin pz.
// f0: explicit spatial force (body coords); B: implicit 6x6.  If a != nullptr also returns the
// implicit-consistent net force (body coords)  sum_c [f0_c - K_c Xc a].
template <int SH> HXD void contact_shape(const DynParams& P, SV v, V3 nb, float pz, SV& f0, SI& B,
                                         const SV* a_true, V3* net_force) {
  const float c_n = P.dn + P.kn * P.dt;
  V3 net = mk(0.f, 0.f, 0.f);
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const V3 r = mk(HXM_CONTACT_PTS[(SH * 8 + k) * 3], HXM_CONTACT_PTS[(SH * 8 + k) * 3 + 1], HXM_CONTACT_PTS[(SH * 8 + k) * 3 + 2]);
    const float pen = -(pz + dot(nb, r));
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
  if (net_force) *net_force = net;
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

struct DynState {
  V3 pos; float quat[4];   // xyzw, body->world
  V3 linvel, angvel;       // world frame
  float q[HX_NJ], qd[HX_NJ];
};

HXD M3 quat_to_mat(const float* q) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  M3 r;
  r.m[0] = 1.f - 2.f * (y * y + z * z); r.m[1] = 2.f * (x * y - z * w); r.m[2] = 2.f * (x * z + y * w);
  r.m[3] = 2.f * (x * y + z * w); r.m[4] = 1.f - 2.f * (x * x + z * z); r.m[5] = 2.f * (y * z - x * w);
  r.m[6] = 2.f * (x * z - y * w); r.m[7] = 2.f * (y * z + x * w); r.m[8] = 1.f - 2.f * (x * x + y * y);
  return r;
}

// One 1 ms substep.  target/kp/kd/tau_lim per joint; mass_scale = base mass / nominal base mass.
// Outputs: tau (the reference's torque, before integration), and if want_forces the net contact force
// per shape body in the world frame (shape order: base, L_thigh, L_toe, R_thigh, R_toe).
HXD void dyn_substep(DynState& S, const DynParams& P, const float* target, const float* kp, const float* kd,
                     const float* tau_lim, float mass_scale, float* tau_out, bool want_forces, V3* shape_force) {
  SV v[HX_NB];
  float cs_c[HX_NB], cs_s[HX_NB];
  M3 Rw[HX_NB];
  float pz[HX_NB];
  // ---- pass 1: kinematics
  {
    M3 R0 = quat_to_mat(S.quat);
    Rw[0] = R0;
    pz[0] = S.pos.z;
    v[0].w = mulT(R0, S.angvel);
    v[0].v = mulT(R0, S.linvel);
  }
  V3 pw[HX_NB];
  pw[0] = S.pos;
  static_for<HX_NJ>([&](auto ic) {
    constexpr int I = decltype(ic)::value + 1;
    constexpr int Pp = BodyC<I>::parent;
    constexpr int K = BodyC<I>::axis;
    float s, c;
    sincosf(S.q[I - 1], &s, &c);
    cs_c[I] = c; cs_s[I] = s;
    const V3 r = BodyC<I>::off();
    const V3 t = v[Pp].v + cross(v[Pp].w, r);
    v[I].w = rotT<K>(c, s, v[Pp].w);
    v[I].v = rotT<K>(c, s, t);
    if (K == 0) v[I].w.x += S.qd[I - 1];
    if (K == 1) v[I].w.y += S.qd[I - 1];
    if (K == 2) v[I].w.z += S.qd[I - 1];
    for (int i = 0; i < 3; ++i) setrow(Rw[I], i, rotT<K>(c, s, row(Rw[Pp], i)));
    pw[I] = pw[Pp] + mul(Rw[Pp], r);
    pz[I] = pw[I].z;
  });

  // ---- pass 2: articulated inertias, leaves to root
  SV U[HX_NB];
  float Dinv[HX_NB], uu[HX_NB];
  SI accI[HX_NB];   // contribution passed to the parent, indexed by the CHILD that produced it
  SV accP[HX_NB];
  SI baseI = BodyC<0>::inertia(mass_scale);
  SV baseP = BodyC<0>::bias(v[0], mass_scale);
  static_for<HX_NJ>([&](auto ic) {
    constexpr int I = HX_NJ - decltype(ic)::value;       // 10..1
    constexpr int K = BodyC<I>::axis;
    constexpr int SH = ShapeOf<I>::value;
    constexpr bool leaf = (I == 5 || I == 10);
    SI IA = BodyC<I>::inertia(1.f);
    SV pA = BodyC<I>::bias(v[I], 1.f);
    if (!leaf) {
      IA.A = IA.A + accI[I + 1].A; IA.H = IA.H + accI[I + 1].H; IA.M = IA.M + accI[I + 1].M;
      pA = pA + accP[I + 1];
    }
    if (SH >= 0) {
      const V3 nb = row(Rw[I], 2);
      SV f0; f0.w = mk(0, 0, 0); f0.v = mk(0, 0, 0);
      SI B; B.A = m3zero(); B.H = m3zero(); B.M = m3zero();
      contact_shape<(SH >= 0 ? SH : 0)>(P, v[I], nb, pz[I], f0, B, nullptr, nullptr);
      IA.A = IA.A + B.A; IA.H = IA.H + B.H; IA.M = IA.M + B.M;
      SV g; g.w = mk(0, 0, 0); g.v = P.gz * nb;
      pA = pA - f0 + mulSI(B, g);
    }
    // joint-space terms
    const float q = S.q[I - 1], qd = S.qd[I - 1];
    const float raw = kp[I - 1] * (target[I - 1] - q) - kd[I - 1] * qd;
    const float tau = fminf(fmaxf(raw, -tau_lim[I - 1]), tau_lim[I - 1]);
    tau_out[I - 1] = tau;
    float beta = (raw == tau) ? P.dt * (kd[I - 1] + P.dt * kp[I - 1]) : 0.f;
    const float c_lim = P.lim_d + P.lim_k * P.dt;
    const float lo_pen = HXM_QLO[I - 1] - q, hi_pen = q - HXM_QHI[I - 1];
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
    U[I] = Ui; Dinv[I] = di; uu[I] = ui;
    // Ia = IA - U U^T / D ; pa = pA + Ia c + U ui / D
    addouter(IA.A, -di, Ui.w, Ui.w);
    addouter(IA.H, -di, Ui.w, Ui.v);
    addouter(IA.M, -di, Ui.v, Ui.v);
    SV cI;   // c_i = v_i x (S qd)
    {
      V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
      cI.w = cross(v[I].w, w2); cI.v = cross(v[I].v, w2);
    }
    SV pa = pA + mulSI(IA, cI);
    pa.w = pa.w + (ui * di) * Ui.w; pa.v = pa.v + (ui * di) * Ui.v;
    // transform to the parent frame:  X^T Ia X,  X^T pa
    const float c = cs_c[I], s = cs_s[I];
    const V3 r = BodyC<I>::off();
    M3 A1 = rotM<K>(c, s, IA.A), H1 = rotM<K>(c, s, IA.H), M1 = rotM<K>(c, s, IA.M);
    M3 G = crossM(r, M1);                       // rx M'
    M3 T1 = crossM(r, transpose(H1));           // rx H'^T
    M3 Kk = crossM(r, transpose(G));            // rx G^T = (G rx^T)^T, symmetric
    SI out;
    out.A = A1 + T1 + transpose(T1) + Kk;
    out.H = H1 + G;
    out.M = M1;
    SV po; po.v = rot<K>(c, s, pa.v); po.w = rot<K>(c, s, pa.w) + cross(r, po.v);
    if (I == 1 || I == 6) {
      baseI.A = baseI.A + out.A; baseI.H = baseI.H + out.H; baseI.M = baseI.M + out.M;
      baseP = baseP + po;
    } else {
      accI[I] = out; accP[I] = po;
    }
  });
  // base: contacts, then solve
  SV a[HX_NB];       // accelerations relative to the gravity field
  V3 g0;
  {
    const V3 nb = row(Rw[0], 2);
    SV f0; f0.w = mk(0, 0, 0); f0.v = mk(0, 0, 0);
    SI B; B.A = m3zero(); B.H = m3zero(); B.M = m3zero();
    contact_shape<0>(P, v[0], nb, pz[0], f0, B, nullptr, nullptr);
    baseI.A = baseI.A + B.A; baseI.H = baseI.H + B.H; baseI.M = baseI.M + B.M;
    SV g; g.w = mk(0, 0, 0); g.v = P.gz * nb;
    g0 = g.v;
    baseP = baseP - f0 + mulSI(B, g);
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
    a[0].w = mk(bm[0], bm[1], bm[2]);
    a[0].v = mk(bm[3], bm[4], bm[5]);
  }
  // ---- pass 3: accelerations, root to leaves
  float qdd[HX_NJ];
  static_for<HX_NJ>([&](auto ic) {
    constexpr int I = decltype(ic)::value + 1;
    constexpr int Pp = BodyC<I>::parent;
    constexpr int K = BodyC<I>::axis;
    const float c = cs_c[I], s = cs_s[I];
    const V3 r = BodyC<I>::off();
    const float qd = S.qd[I - 1];
    SV ai;
    ai.w = rotT<K>(c, s, a[Pp].w);
    ai.v = rotT<K>(c, s, a[Pp].v + cross(a[Pp].w, r));
    V3 w2 = mk(K == 0 ? qd : 0.f, K == 1 ? qd : 0.f, K == 2 ? qd : 0.f);
    ai.w = ai.w + cross(v[I].w, w2);
    ai.v = ai.v + cross(v[I].v, w2);
    const float dd = Dinv[I] * (uu[I] - (dot(U[I].w, ai.w) + dot(U[I].v, ai.v)));
    qdd[I - 1] = dd;
    if (K == 0) ai.w.x += dd;
    if (K == 1) ai.w.y += dd;
    if (K == 2) ai.w.z += dd;
    a[I] = ai;
  });
  // ---- net contact forces (implicit-consistent), only when asked (last substep of an env step)
  if (want_forces) {
    static_for<HX_NB>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int SH = ShapeOf<I>::value;
      if (SH >= 0) {
        const V3 nb = row(Rw[I], 2);
        SV at = a[I]; at.v = at.v + P.gz * nb;     // true spatial acceleration
        SV f0d; SI Bd; V3 net;
        contact_shape<(SH >= 0 ? SH : 0)>(P, v[I], nb, pz[I], f0d, Bd, &at, &net);
        shape_force[SH >= 0 ? SH : 0] = mul(Rw[I], net);
      }
    });
  }
  // ---- integrate (semi-implicit Euler)
  {
    const V3 a_ang = a[0].w;
    const V3 a_lin = a[0].v + g0 + cross(v[0].w, v[0].v);
    S.angvel = S.angvel + P.dt * mul(Rw[0], a_ang);
    S.linvel = S.linvel + P.dt * mul(Rw[0], a_lin);
    for (int j = 0; j < HX_NJ; ++j) {
      const float nqd = S.qd[j] + P.dt * qdd[j];
      S.qd[j] = fminf(fmaxf(nqd, -HXM_VMAX[j]), HXM_VMAX[j]);
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

// Forward kinematics for the observation/reward glue: world position, linear and angular velocity of the
// body origins of bodies 4,5,9,10 (calves = "knees", toes = "feet"; reference hector_config.py:31-32).
struct BodyOut { V3 pos, linvel, angvel; float quat[4]; };
HXD void dyn_body_states(const DynState& S, BodyOut* out /*[4]: L_calf, L_toe, R_calf, R_toe*/) {
  SV v[HX_NB]; M3 Rw[HX_NB]; V3 pw[HX_NB];
  Rw[0] = quat_to_mat(S.quat);
  pw[0] = S.pos;
  v[0].w = mulT(Rw[0], S.angvel);
  v[0].v = mulT(Rw[0], S.linvel);
  static_for<HX_NJ>([&](auto ic) {
    constexpr int I = decltype(ic)::value + 1;
    constexpr int Pp = BodyC<I>::parent;
    constexpr int K = BodyC<I>::axis;
    float s, c;
    sincosf(S.q[I - 1], &s, &c);
    const V3 r = BodyC<I>::off();
    const V3 t = v[Pp].v + cross(v[Pp].w, r);
    v[I].w = rotT<K>(c, s, v[Pp].w);
    v[I].v = rotT<K>(c, s, t);
    if (K == 0) v[I].w.x += S.qd[I - 1];
    if (K == 1) v[I].w.y += S.qd[I - 1];
    if (K == 2) v[I].w.z += S.qd[I - 1];
    for (int i = 0; i < 3; ++i) setrow(Rw[I], i, rotT<K>(c, s, row(Rw[Pp], i)));
    pw[I] = pw[Pp] + mul(Rw[Pp], r);
    constexpr int slot = (I == 4) ? 0 : (I == 5) ? 1 : (I == 9) ? 2 : (I == 10) ? 3 : -1;
    if (slot >= 0) {
      BodyOut& o = out[slot >= 0 ? slot : 0];
      o.pos = pw[I];
      o.linvel = mul(Rw[I], v[I].v);
      o.angvel = mul(Rw[I], v[I].w);
      // rotation matrix -> xyzw quaternion (w >= 0), largest-component branch
      const M3& R = Rw[I];
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
      o.quat[0] = sg * qx; o.quat[1] = sg * qy; o.quat[2] = sg * qz; o.quat[3] = sg * qw;
    }
  });
}
