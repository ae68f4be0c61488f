// hx_math.h -- small fixed-size linear algebra and the lane primitives shared by the simulator's device code and its
// host (CPU) build.  Single source: hipcc compiles it for gfx950, g++ compiles the same text for the OpenMP host build
// (oracle/host/, the cpu_baseline and sanitizer target -- never part of the product path).
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HXD __host__ __device__ __forceinline__
#define HX_TABLE __device__ static const
#else
#include <cmath>
#include <cstdint>
#include <cstring>
using std::memcpy;
#if defined(HX_NO_FORCE_INLINE)      /* sanitizer build: separate functions compile in seconds instead of minutes */
#define HXD inline
#else
#define HXD inline __attribute__((always_inline))
#endif
#define HX_TABLE static const
#endif
#include <utility>

// ---- lane primitives.  Device: EIGHT lanes per robot -- lanes 8e .. 8e+3 are the left body side of robot e, lanes
// 8e+4 .. 8e+7 the right side.  The four lanes of a side (a DPP quad) run the same articulated-body recursion redundantly
// and SHARE the side's contact points (lane k takes points k, k+4, ...; hx_qsum adds the four partial sums, after which
// the quad's lanes agree bitwise).  The other side's value comes through a DPP half-row mirror (lane i <-> 7 - i of each
// group of eight: a left lane reads a right lane, all of which hold the same number).  Both are single VALU instructions,
// no LDS round trip.  Host: one robot at a time, both sides in sequence -- the drivers never call hx_xchg there and the
// quad operations are the identity.
#define HX_LANES_PER_ROBOT 8
#define HX_LANES_PER_SIDE 4
#if defined(__HIPCC__)
__device__ __forceinline__ float hx_dpp_quad_xor1(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float hx_dpp_quad_xor2(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float hx_xchg(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true)); }           // row_half_mirror
#endif
// sum / maximum over the four lanes of a body side; every lane of the quad receives the same result (the additions are
// arranged so: (a + b) + (c + d) with both operand orders commuting bitwise)
HXD float hx_qsum(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  x = x + hx_dpp_quad_xor1(x);
  return x + hx_dpp_quad_xor2(x);
#else
  return x;
#endif
}
HXD float hx_qmax(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  x = fmaxf(x, hx_dpp_quad_xor1(x));
  return fmaxf(x, hx_dpp_quad_xor2(x));
#else
  return x;
#endif
}
HXD bool hx_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __any((int)p) != 0;
#else
  return p;
#endif
}
HXD uint32_t hx_fbits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(x);
#else
  uint32_t u; memcpy(&u, &x, 4); return u;
#endif
}
HXD uint32_t hx_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}
// statistics accumulators shared by all robots (host build: OpenMP threads)
HXD float hx_atomic_add(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  float old;
#pragma omp atomic capture
  { old = *p; *p += v; }
  return old;
#endif
}
HXD int hx_atomic_add(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  int old;
#pragma omp atomic capture
  { old = *p; *p += v; }
  return old;
#endif
}
// order LDS traffic between the lanes of one wave (values written by one lane, read by its neighbours): the wave runs in
// lockstep, so waiting for the outstanding LDS operations is all it takes -- no workgroup barrier
HXD void hx_lds_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
HXD int hx_imin(int a, int b) { return a < b ? a : b; }
HXD int hx_imax(int a, int b) { return a > b ? a : b; }

struct V3 { float x, y, z; };
HXD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
HXD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
HXD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
HXD V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
HXD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
HXD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HXD V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HXD float get(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
HXD V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

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
HXD M3 matmul(const M3& a, const M3& b) { M3 r; for (int j = 0; j < 3; ++j) setcol(r, j, mul(a, col(b, j))); return r; }
HXD M3 ld9(const float* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
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
// E A E^T for a general rotation E
HXD M3 simM(const M3& E, const M3& a) { return matmul(matmul(E, a), transpose(E)); }

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
HXD SV sv0() { SV r; r.w = mk(0.f, 0.f, 0.f); r.v = mk(0.f, 0.f, 0.f); return r; }
struct SI { M3 A, H, M; };   // 6x6 symmetric [[A,H],[H^T,M]]
HXD SI si0() { SI r; r.A = m3zero(); r.H = m3zero(); r.M = m3zero(); return r; }
HXD void siadd(SI& a, const SI& b) { a.A = a.A + b.A; a.H = a.H + b.H; a.M = a.M + b.M; }
HXD SV mulSI(const SI& I, SV a) { SV f; f.w = mul(I.A, a.w) + mul(I.H, a.v); f.v = mulT(I.H, a.w) + mul(I.M, a.v); return f; }

template <typename F, int... Is> HXD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> HXD void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

HXD M3 quat_to_mat(const float* q) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  M3 r;
  r.m[0] = 1.f - 2.f * (y * y + z * z); r.m[1] = 2.f * (x * y - z * w); r.m[2] = 2.f * (x * z + y * w);
  r.m[3] = 2.f * (x * y + z * w); r.m[4] = 1.f - 2.f * (x * x + z * z); r.m[5] = 2.f * (y * z - x * w);
  r.m[6] = 2.f * (x * z - y * w); r.m[7] = 2.f * (y * z + x * w); r.m[8] = 1.f - 2.f * (x * x + y * y);
  return r;
}
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
