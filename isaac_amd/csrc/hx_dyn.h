// hx_dyn.h -- articulated-body dynamics of the hector family of bipeds (fp32), single source for the gfx950 kernels
// and the host build (hx_math.h explains the split).
//
// Device layout: EIGHT lanes per robot, four per body side (hx_math.h).  A side's four lanes run the side's recursion
// redundantly and share its contact points; both sides carry the floating base redundantly.  Every robot of the family is a base with kinematic chains hanging off it (hector: one
// 5-joint leg per side; hector_full: a leg and a 4-joint arm per side; XBot-L: one 6-joint leg per side), so
// Featherstone's articulated-body algorithm splits cleanly per side: a lane runs the leaf-to-root recursion of its own
// chains and the two sides' contributions to the base (6x6 articulated inertia + bias force, 33 floats) are summed with
// one cross-side exchange (DPP half-row mirror, hx_xchg).  Per-side constants (joint offsets, inertias, limits, PD gains,
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
// Self-collision (asset.self_collisions = 0, humanoid_config.py:66; hector's configs disable it): the two legs of a biped can
// only meet foot against foot and knee against knee, so a model lists NSELF side-local bodies that carry one sphere each --
// SELF_AT_SHAPE: centred on the body's collision shape (its bounding-sphere centre) or on the body origin (the joint) -- and
// body k of the left side collides with body k of the right side.  PhysX collides the links' convex hulls pairwise; the
// spheres are this model's proxy for the pairs that can touch.
struct ModelHector : HXM_Hector {
  static constexpr bool ARMS = false, XBOT = false;
  static constexpr int NSELF = 0;
  static constexpr int SELF_BODY[1] = {0}; static constexpr bool SELF_AT_SHAPE[1] = {false}; static constexpr float SELF_RADIUS[1] = {0.f};
  static constexpr int KNEE = 3, FOOT = 4;       // side-local bodies named by asset.knee_name / foot_name (hector_config.py:31-32)
  HXD static const float* side_table() { return HXM_SIDE; }
  HXD static const float* base_table() { return HXM_BASE; }
  static constexpr float MASS0 = HXM_MASS0;
};
struct ModelFull : HXM_Full {             // hector with arms (task hector_full): leg bodies 0-4, arm bodies 5-8 per side
  static constexpr bool ARMS = true, XBOT = false;
  static constexpr int NSELF = 0;
  static constexpr int SELF_BODY[1] = {0}; static constexpr bool SELF_AT_SHAPE[1] = {false}; static constexpr float SELF_RADIUS[1] = {0.f};
  static constexpr int KNEE = 3, FOOT = 4;
  HXD static const float* side_table() { return HXF_SIDE; }
  HXD static const float* base_table() { return HXF_BASE; }
  static constexpr float MASS0 = HXF_MASS0;
};
struct ModelXBot : HXM_XBot {             // XBot-L (task humanoid_ppo): roll, yaw, pitch, knee, ankle pitch, ankle roll
  static constexpr bool ARMS = false, XBOT = true;
  static constexpr int KNEE = 3, FOOT = 5;       // 'knee' / 'ankle_roll' (humanoid_config.py:64-65)
  // knee against knee: 6 cm spheres on the knee joints (origin of knee_link); foot against foot: 5 cm spheres (half the sole's
  // width) on the centre of the ankle_roll_link hull
  static constexpr int NSELF = 2;
  static constexpr int SELF_BODY[2] = {3, 5}; static constexpr bool SELF_AT_SHAPE[2] = {false, true}; static constexpr float SELF_RADIUS[2] = {0.06f, 0.05f};
  HXD static const float* side_table() { return HXX_SIDE; }
  HXD static const float* base_table() { return HXX_BASE; }
  static constexpr float MASS0 = HXX_MASS0;
};
template <class M> struct ModelInfo {
  static constexpr int nshape() { int n = 0; for (int k = 0; k < M::NL; ++k) n += M::NPTS[k] > 0 ? 1 : 0; return n; }
  static constexpr int slot(int B) { int n = 0; for (int k = 0; k < B; ++k) n += M::NPTS[k] > 0 ? 1 : 0; return M::NPTS[B] > 0 ? n : -1; }
  static constexpr int chain_start_of(int B) { int s = 0; for (int c = 0; c < M::NCH; ++c) if (B >= M::CH_START[c] && B < M::CH_START[c] + M::CH_LEN[c]) s = M::CH_START[c]; return s; }
  static constexpr int NSHAPE = nshape();
  static constexpr int NSLOT = NSHAPE + 1;                 // contact-buffer slots of a lane: its shape bodies, then the base
  static constexpr int NENT = NSHAPE + M::BASE_NSUB;       // entries of the lane's contact loop: shape bodies, then its base sub-shapes
  static constexpr int PD_OFF = 2 * M::SIDE_STRIDE + M::BASE_FLOATS;
  static constexpr int ENT_OFF = PD_OFF + 2 * M::NL * 4;   // [side][NENT][3] ints: float offset of the shape block, points, buffer slot
  static constexpr int LDS_FLOATS = ENT_OFF + 2 * NENT * 3;
};

// Terrain: each robot keeps a HX_PATCH x HX_PATCH window of the height grid (metres, fp32) in LDS, centred on its
// base at the start of the env step; `patch == nullptr` selects the ground plane z = 0.
#define HX_PATCH 16
#define HX_POOL 8
#define HX_PATCH_LD (HX_PATCH * HX_PATCH + 1)      /* per-robot LDS strides: odd, so that the 32 robots of a wave reading the */
#define HX_POOL_LD (HX_POOL * HX_POOL + 1)          /* same cell of their own windows hit 32 different banks */
struct DynParams {
  float dt, inv_dt, gz, kn, dn, veps, lim_k, lim_d, mu;
  // the contact inputs the reference states for PhysX (hector_config.py:113-117), in this penalty model's terms:
  float vdep;             // max_depenetration_velocity: the spring part of a point's normal force is capped at c_n * vdep, the force
                          // at which a point pushes itself out at vdep -- a deep penetration (a toe driven into a riser) is
                          // corrected at that speed instead of being thrown out by k * depth; <= 0: no cap
  float coff;             // contact_offset: a point closer than this to the surface is a contact already; while it is still
                          // outside, the damper acts on the part of its approach speed that would carry it through the surface
                          // within this substep (vn + gap / dt < 0) -- the penalty form of PhysX's speculative contact
  float roff;             // rest_offset: distance at which shapes come to rest (added to the penetration)
  int self_on;            // 1: the model's self-collision pairs are active (asset.self_collisions = 0)
  int tflags;             // ablation switches of the trimesh walls (hx_sim_set_terrain_options): 1 = cliff cells keep their
                          // ramp (no flattening), 2 = no sideways wall contact
  const float* patch;     // LDS, [HX_PATCH][HX_PATCH] row-major (row = x index), or nullptr
  float px0, py0;         // world x / y of patch node (0, 0)
  float inv_hs;           // 1 / horizontal_scale
  const float* pool;      // LDS, [HX_POOL][HX_POOL]: pool[I][J] = highest node among nodes 2I-2 .. 2I+4 x 2J-2 .. 2J+4 of the
                          // patch -- an upper bound of the surface under any point whose cell (i, j) has (i/2, j/2) = (I, J),
                          // and under a whole shape of bounding radius <= 0.2 m centred over such a cell
  const float* poolw;     // LDS, same shape: 1 if two neighbouring nodes of that region differ by more than `wall` (only then
                          // can a point over cell (I, J) meet a wall or a cliff cell), else 0
  float wall;             // height difference between grid neighbours beyond which the trimesh has a vertical wall
                          // (slope_treshold * horizontal_scale, reference utils/terrain.py:70-73); 0 = no walls (heightfield)
  long long* prof;        // measurement builds (-DHX_STEP_PROF): per-wave cycle counters in LDS, else unused
  int pt0, ptstep;        // this lane's share of a shape's points: pt0, pt0 + ptstep, ... (device: lane & 3, 4; host: 0, 1)
};
// Phase timers of the env-step kernel (tools/step_prof.py): lane 0 of a wave adds the shader-clock cycles since the
// previous mark to counter `id`.  Compiled out unless -DHX_STEP_PROF.
#if defined(HX_STEP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define HX_T(prof, id) do { if ((prof) != nullptr && threadIdx.x == 0) { const long long t_ = clock64(); (prof)[id] += t_ - (prof)[15]; (prof)[15] = t_; } } while (0)
#else
#define HX_T(prof, id) do { } while (0)
#endif

// stage the per-side tables, the base table, the PD constants of both sides (kp[ND], kd[ND], tau_lim[ND], default_pos[ND]
// in DoF order = left side then right side) and the contact-loop entry tables into LDS; call with all threads, then
// synchronise
template <class M> HXD void dyn_stage_constants(float* lds, int tid, int nthreads, const float* kp, const float* kd, const float* tl, const float* q0) {
  using MI = ModelInfo<M>;
  for (int i = tid; i < 2 * M::SIDE_STRIDE; i += nthreads) lds[i] = M::side_table()[i];
  for (int i = tid; i < M::BASE_FLOATS; i += nthreads) lds[2 * M::SIDE_STRIDE + i] = M::base_table()[i];
  float* pd = lds + MI::PD_OFF;
  for (int i = tid; i < 2 * M::NL; i += nthreads) { pd[4 * i] = kp[i]; pd[4 * i + 1] = kd[i]; pd[4 * i + 2] = tl[i]; pd[4 * i + 3] = q0[i]; }
  int* ent = reinterpret_cast<int*>(lds + MI::ENT_OFF);
  for (int i = tid; i < 2 * MI::NENT; i += nthreads) {
    const int side = i / MI::NENT, e = i % MI::NENT;
    int off = 0, np = 0, slot = 0;
    if (e < MI::NSHAPE) {
      int b = 0;
      for (int k = 0; k < M::NL; ++k) if (MI::slot(k) == e) b = k;
      off = side * M::SIDE_STRIDE + M::PTS_OFF[b]; np = M::NPTS[b]; slot = e;
    } else {
      off = 2 * M::SIDE_STRIDE + (side * M::BASE_NSUB + (e - MI::NSHAPE)) * (4 + 3 * M::BASE_NP); np = M::BASE_NP; slot = MI::NSHAPE;
    }
    ent[3 * i] = off; ent[3 * i + 1] = np; ent[3 * i + 2] = slot;
  }
}

// one side's view of the constants
template <class M> struct SideConst {
  const float* lds;    // all staged constants
  const float* t;      // side table
  const float* bt;     // base table: 2 x BASE_NSUB sub-shapes [sphere 4][points 3 * BASE_NP], then inertia 6, h 3, mass
  const float* pd;     // [NL][4] kp, kd, tau_lim, default position of this side's joints
  const int* ent;      // [NENT][3] contact-loop entries of this side
  const float* bsub;   // this lane's BASE_NSUB base sub-shapes
  HXD void bind(const float* l, int side) {
    lds = l; t = l + side * M::SIDE_STRIDE; bt = l + 2 * M::SIDE_STRIDE; pd = l + ModelInfo<M>::PD_OFF + side * M::NL * 4;
    ent = reinterpret_cast<const int*>(l + ModelInfo<M>::ENT_OFF) + side * ModelInfo<M>::NENT * 3;
    bsub = bt + side * M::BASE_NSUB * (4 + 3 * M::BASE_NP);
  }
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
  static constexpr int BASE_IO = 2 * M::BASE_NSUB * (4 + 3 * M::BASE_NP);
  HXD V3 base_h() const { return ld3(bt + BASE_IO + 6); }
  HXD float base_mass() const { return bt[BASE_IO + 9]; }
  HXD SI base_inertia(float s) const { return spatial(bt + BASE_IO, base_h(), base_mass(), s); }
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
#ifndef HX_PT_GROUP
#define HX_PT_GROUP 2      // points of a shape that a lane takes through the contact stages together
#endif
struct TerrainCell { float h00, h01, h10, h11, fu, fw; };
// the four corner heights of the cell under (u, w) and the position inside it: the LDS reads of a query, apart from its arithmetic
HXD TerrainCell terrain_fetch(const DynParams& P, float u, float w) {
  const int i = hx_imin(hx_imax((int)floorf(u), 0), HX_PATCH - 2), j = hx_imin(hx_imax((int)floorf(w), 0), HX_PATCH - 2);
  TerrainCell c;
  c.fu = fminf(fmaxf(u - (float)i, 0.f), 1.f); c.fw = fminf(fmaxf(w - (float)j, 0.f), 1.f);
  const float* g = P.patch + i * HX_PATCH + j;
  c.h00 = g[0]; c.h01 = g[1]; c.h10 = g[HX_PATCH]; c.h11 = g[HX_PATCH + 1];
  return c;
}
HXD float terrain_eval(const DynParams& P, const TerrainCell& c, V3& nw, bool walls) {
  float h00 = c.h00, h01 = c.h01, h10 = c.h10, h11 = c.h11;
  const float fu = c.fu, fw = c.fw;
  if (hx_any(walls) && !(P.tflags & 1)) {
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
HXD float terrain_query(const DynParams& P, float u, float w, V3& nw, bool walls) {
  const TerrainCell c = terrain_fetch(P, u, w);
  return terrain_eval(P, c, nw, walls);
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

// index of the pool entry over patch coordinates (u, w): see DynParams::pool
HXD int terrain_pool_index(float u, float w) {
  const int i = hx_imin(hx_imax((int)floorf(u), 0), HX_PATCH - 2), j = hx_imin(hx_imax((int)floorf(w), 0), HX_PATCH - 2);
  return (i >> 1) * HX_POOL + (j >> 1);
}
// Per-lane contact buffer.  The contact phase of a substep is ONE runtime loop over the lane's collision shapes (a single
// instance of the point loop in the instruction stream instead of one inlined copy per shape); it takes each shape's body
// state from this buffer and leaves the accumulated explicit force f0 and implicit 6x6 term B there for the articulated-
// inertia pass, and for the net contact forces read after the accelerations are known.  Device: LDS, [slot][field][lane];
// host: a local array (stride 1).
#define HX_CB_V 0        /* in : spatial velocity of the body, body coords (w, v)            6 */
#define HX_CB_R 6        /* in : body -> world rotation, row-major                            9 */
#define HX_CB_P 15       /* in : world position of the body origin                            3 */
#define HX_CB_F 18       /* out: f0 (w, v)                                                     6 */
#define HX_CB_A 24       /* out: B.A symmetric (xx xy xz yy yz zz)                             6 */
#define HX_CB_H 30       /* out: B.H row-major                                                 9 */
#define HX_CB_M 39       /* out: B.M symmetric                                                 6 */
#define HX_CB_FIELDS 45
struct ContactBuf {
  float* base; int stride;
  HXD float& at(int slot, int field) const { return base[(slot * HX_CB_FIELDS + field) * stride]; }
  HXD void put_body(int slot, const SV& v, const M3& Rb, V3 pb) const {
    at(slot, HX_CB_V) = v.w.x; at(slot, HX_CB_V + 1) = v.w.y; at(slot, HX_CB_V + 2) = v.w.z;
    at(slot, HX_CB_V + 3) = v.v.x; at(slot, HX_CB_V + 4) = v.v.y; at(slot, HX_CB_V + 5) = v.v.z;
    for (int i = 0; i < 9; ++i) at(slot, HX_CB_R + i) = Rb.m[i];
    at(slot, HX_CB_P) = pb.x; at(slot, HX_CB_P + 1) = pb.y; at(slot, HX_CB_P + 2) = pb.z;
  }
  HXD M3 rot_of(int slot) const { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = at(slot, HX_CB_R + i); return r; }
  // f0 and B of a slot added to (IA, pA):  IA += B ;  pA += -f0 + B g   (g = gravity field in body coords, linear part only)
  HXD void add_to(int slot, SI& IA, SV& pA, V3 g) const {
    SV f0; f0.w = mk(at(slot, HX_CB_F), at(slot, HX_CB_F + 1), at(slot, HX_CB_F + 2)); f0.v = mk(at(slot, HX_CB_F + 3), at(slot, HX_CB_F + 4), at(slot, HX_CB_F + 5));
    SI B;
    const float axx = at(slot, HX_CB_A), axy = at(slot, HX_CB_A + 1), axz = at(slot, HX_CB_A + 2), ayy = at(slot, HX_CB_A + 3), ayz = at(slot, HX_CB_A + 4), azz = at(slot, HX_CB_A + 5);
    B.A.m[0] = axx; B.A.m[1] = axy; B.A.m[2] = axz; B.A.m[3] = axy; B.A.m[4] = ayy; B.A.m[5] = ayz; B.A.m[6] = axz; B.A.m[7] = ayz; B.A.m[8] = azz;
    for (int i = 0; i < 9; ++i) B.H.m[i] = at(slot, HX_CB_H + i);
    const float mxx = at(slot, HX_CB_M), mxy = at(slot, HX_CB_M + 1), mxz = at(slot, HX_CB_M + 2), myy = at(slot, HX_CB_M + 3), myz = at(slot, HX_CB_M + 4), mzz = at(slot, HX_CB_M + 5);
    B.M.m[0] = mxx; B.M.m[1] = mxy; B.M.m[2] = mxz; B.M.m[3] = mxy; B.M.m[4] = myy; B.M.m[5] = myz; B.M.m[6] = mxz; B.M.m[7] = myz; B.M.m[8] = mzz;
    siadd(IA, B);
    SV gg; gg.w = mk(0.f, 0.f, 0.f); gg.v = g;
    pA = pA - f0 + mulSI(B, gg);
  }
  // implicit-consistent net force of a slot (body coords):  sum_c [f_c - K_c Xc a] = f0.v - (B a).v = f0.v - (H^T a.w + M a.v)
  HXD V3 net_force(int slot, const SV& a) const {
    M3 H; for (int i = 0; i < 9; ++i) H.m[i] = at(slot, HX_CB_H + i);
    const float mxx = at(slot, HX_CB_M), mxy = at(slot, HX_CB_M + 1), mxz = at(slot, HX_CB_M + 2), myy = at(slot, HX_CB_M + 3), myz = at(slot, HX_CB_M + 4), mzz = at(slot, HX_CB_M + 5);
    const V3 ma = mk(mxx * a.v.x + mxy * a.v.y + mxz * a.v.z, mxy * a.v.x + myy * a.v.y + myz * a.v.z, mxz * a.v.x + myz * a.v.y + mzz * a.v.z);
    return mk(at(slot, HX_CB_F + 3), at(slot, HX_CB_F + 4), at(slot, HX_CB_F + 5)) - (mulT(H, a.w) + ma);
  }
};

// Bounding-sphere test of a shape (`shp`: centre xyz + radius, then the points) on a body with world pose (Rb, pb): can any
// lane's shape touch at all?  Nothing of it can while its lowest possible point is above the highest ground in reach: 0 on
// the plane; on terrain the pooled maximum around the sphere's centre (radius <= 0.2 m: one pool entry; wider shapes: the
// 3 x 3 entries around, reach 0.4 m).  Returns lowest point minus bound, per lane: negative = may touch.  Runs in the
// kinematics pass, where the pose is in registers, so that the contact loop only ever visits shapes that may touch; the
// tests of a pass are evaluated without branches in between, so their LDS latencies overlap with the recursion.
HXD float shape_gap(const DynParams& P, const float* shp, const M3& Rb, V3 pb) {
  const V3 c = ld3(shp);
  const float zlow = pb.z + dot(row(Rb, 2), c) - shp[3];
  float bound = 0.f;
  if (P.patch != nullptr) {
    const float u = (pb.x + dot(row(Rb, 0), c) - P.px0) * P.inv_hs, w = (pb.y + dot(row(Rb, 1), c) - P.py0) * P.inv_hs;
    const int pi = terrain_pool_index(u, w);
    bound = P.pool[pi];
    if (shp[3] > 0.2f) {            // uniform branch
      const int i = pi / HX_POOL, j = pi % HX_POOL;
      for (int a = hx_imax(i - 1, 0); a <= hx_imin(i + 1, HX_POOL - 1); ++a)
        for (int b = hx_imax(j - 1, 0); b <= hx_imin(j + 1, HX_POOL - 1); ++b) bound = fmaxf(bound, P.pool[a * HX_POOL + b]);
    }
  }
  return zlow - bound - P.coff;
}
// Second stage of that test, for a shape the sphere could not exclude: the geometric part of contact_shape's own activity test
// (a point takes part only if its penetration exceeds -contact offset), for this lane's points, with the body pose still in
// registers and the same arithmetic as contact_shape -- position, cell, triangle plane, penetration along the normal.  The pooled
// bound of the first stage is the highest ground within 0.3 m: next to a stair riser, a kerb or a pit wall it flags every shape
// of the robot, substep after substep, and every such visit walks its points down to this very test before it finds nothing
// (profiles/r04_al_env_waves.txt: 2.4 visits per substep, 1.0 of them with a contact; a wave with such a robot: 6 and 1.3).
// Returns < 0 if one of the lane's points may be active; HX_GAP_MARGIN keeps the decision on the visiting side of rounding.
#define HX_GAP_MARGIN 1e-4f
template <int NP>
HXD float shape_gap_points(const DynParams& P, const float* shp, const M3& Rb, V3 pb) {
  const float* pts = shp + 4;
  const V3 zb = row(Rb, 2);
  float gap = 1.f;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NJ = (NP + HX_LANES_PER_SIDE - 1) / HX_LANES_PER_SIDE;
#else
  constexpr int NJ = NP;
#endif
  V3 r[NJ]; bool valid[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { const int k = P.pt0 + j * P.ptstep; valid[j] = k < NP; r[j] = ld3(pts + 3 * (valid[j] ? k : P.pt0)); }
  float z[NJ], u[NJ], w[NJ], cliff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    z[j] = pb.z + dot(zb, r[j]);
    u[j] = 0.f; w[j] = 0.f; cliff[j] = 0.f;
    if (P.patch != nullptr) {
      u[j] = (pb.x + dot(row(Rb, 0), r[j]) - P.px0) * P.inv_hs; w[j] = (pb.y + dot(row(Rb, 1), r[j]) - P.py0) * P.inv_hs;
      cliff[j] = P.poolw[terrain_pool_index(u[j], w[j])];
    }
  }
  TerrainCell cell[NJ];
  if (P.patch != nullptr) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) cell[j] = terrain_fetch(P, u[j], w[j]);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float pen = -z[j];
    if (P.patch != nullptr) { V3 nw; const float h = terrain_eval(P, cell[j], nw, cliff[j] != 0.f); pen = (h - z[j]) * nw.z; }
    if (valid[j]) gap = fminf(gap, -(pen + P.roff + P.coff) - HX_GAP_MARGIN);
  }
  return gap;
}

// Contact terms of one shape (`shp`: bounding sphere centre xyz + radius, then npts xyz triples; LDS): the body state
// comes from slot `in` of the buffer, explicit force and implicit 6x6 are ACCUMULATED into slot `out` (several base
// sub-shapes share one output).  Plane: normal = world z, penetration = -z.  Terrain: normal of the triangle under the
// point, penetration = distance to that triangle's plane; or the wall the point went through (wall_push).
HXD bool contact_shape(const DynParams& P, const float* shp, int npts, const ContactBuf& cb, int in, int out, bool accumulate) {
  const V3 pb = mk(cb.at(in, HX_CB_P), cb.at(in, HX_CB_P + 1), cb.at(in, HX_CB_P + 2));
  const M3 Rb = cb.rot_of(in);
  const V3 zb = row(Rb, 2);          // world z in body coords
  SV v; v.w = mk(cb.at(in, HX_CB_V), cb.at(in, HX_CB_V + 1), cb.at(in, HX_CB_V + 2)); v.v = mk(cb.at(in, HX_CB_V + 3), cb.at(in, HX_CB_V + 4), cb.at(in, HX_CB_V + 5));
  const float c_n = P.dn + P.kn * P.dt;
  const float* pts = shp + 4;
  SV f0 = sv0();
  float Axx = 0.f, Axy = 0.f, Axz = 0.f, Ayy = 0.f, Ayz = 0.f, Azz = 0.f, Mxx = 0.f, Mxy = 0.f, Mxz = 0.f, Myy = 0.f, Myz = 0.f, Mzz = 0.f;
  M3 H = m3zero();
  float any_on = 0.f;
  // The lane's points in groups of HX_PT_GROUP (device: a lane owns <= 3 of a shape's points, i.e. one or two groups), every group in
  // STAGES that each issue the LDS reads of all its points before anything waits for one of them: point coordinates -> pooled
  // bounds and cliff flags -> one ballot -> corner heights -> forces.  One point at a time (round 3) made every link of that chain
  // an exposed LDS round trip per point: 9 per foot shape and lane instead of 3 (profiles/r03_n_env_step_stalls.txt).  The
  // ballots only skip work nobody needs, so the sums -- taken in point order as before -- are the same.
#pragma unroll 1
  for (int k0 = P.pt0; k0 < npts; k0 += HX_PT_GROUP * P.ptstep) {
    V3 r[HX_PT_GROUP]; float z[HX_PT_GROUP], u[HX_PT_GROUP], w[HX_PT_GROUP]; int pi[HX_PT_GROUP]; bool valid[HX_PT_GROUP], near_[HX_PT_GROUP];
#pragma unroll
    for (int j = 0; j < HX_PT_GROUP; ++j) {
      const int k = k0 + j * P.ptstep;
      valid[j] = k < npts;
      r[j] = ld3(pts + 3 * (valid[j] ? k : k0));
    }
#pragma unroll
    for (int j = 0; j < HX_PT_GROUP; ++j) {
      z[j] = pb.z + dot(zb, r[j]);
      u[j] = 0.f; w[j] = 0.f; pi[j] = 0;
      if (P.patch != nullptr) {
        u[j] = (pb.x + dot(row(Rb, 0), r[j]) - P.px0) * P.inv_hs; w[j] = (pb.y + dot(row(Rb, 1), r[j]) - P.py0) * P.inv_hs;
        pi[j] = terrain_pool_index(u[j], w[j]);
      }
    }
    float bound[HX_PT_GROUP], cliff[HX_PT_GROUP];
    bool any_near = false;
#pragma unroll
    for (int j = 0; j < HX_PT_GROUP; ++j) { bound[j] = 0.f; cliff[j] = 0.f; if (P.patch != nullptr) { bound[j] = P.pool[pi[j]]; cliff[j] = P.poolw[pi[j]]; } }
#pragma unroll
    for (int j = 0; j < HX_PT_GROUP; ++j) { near_[j] = valid[j] && (z[j] < bound[j] + P.coff); any_near = any_near || near_[j]; }
    if (!hx_any(any_near)) continue;
#if defined(HX_STEP_PROF) && defined(__HIP_DEVICE_COMPILE__)
    if (P.prof != nullptr && threadIdx.x == 0) P.prof[13] += 1;      // [13] point groups that passed the pooled per-point test (any side)
#endif
    TerrainCell cell[HX_PT_GROUP];
    if (P.patch != nullptr) {
#pragma unroll
      for (int j = 0; j < HX_PT_GROUP; ++j) cell[j] = terrain_fetch(P, u[j], w[j]);
    }
#pragma unroll
    for (int j = 0; j < HX_PT_GROUP; ++j) {
      if (!hx_any(near_[j])) continue;            // nobody's point j is inside its bound (uniform)
      V3 nb = zb;
      float pen = -z[j];
      if (P.patch != nullptr) {
        const bool walls = cliff[j] != 0.f;         // never set without walls (P.wall = 0)
        V3 nw;
        const float h = terrain_eval(P, cell[j], nw, walls);
        pen = (h - z[j]) * nw.z;
        if (hx_any(valid[j] && walls && pen > 0.f) && !(P.tflags & 2)) {
          float wp = pen; V3 wn = nw;
          if (valid[j] && walls && pen > 0.f && wall_push(P, u[j], w[j], z[j], wp, wn)) { pen = wp; nw = wn; }
        }
        nb = mulT(Rb, nw);
      }
      const V3 rj = r[j];
      const V3 vp = v.v + cross(v.w, rj);
      const float vn = dot(vp, nb);
      // normal force = capped spring on the penetration - damper on the closing speed; outside the surface but inside the
      // contact offset only the speed in excess of gap / dt is damped (DynParams::coff, ::vdep)
      pen += P.roff;
      const float spring = P.kn * fmaxf(pen, 0.f);
      const float fn0 = ((P.vdep > 0.f) ? fminf(spring, c_n * P.vdep) : spring) - c_n * (vn + fmaxf(-pen, 0.f) * P.inv_dt);
      const bool act = valid[j] && (pen > -P.coff) && (fn0 > 0.f);      // as before: a point visited because SOME lane's is near takes part on its own merits
      if (!hx_any(act)) continue;
      const V3 vt = vp - vn * nb;
      const float vtn = sqrtf(dot(vt, vt));
      const float c_t = P.mu * fn0 / fmaxf(vtn, P.veps);
      const float on = act ? 1.f : 0.f;
      any_on = fmaxf(any_on, on);
      const V3 f = on * (fn0 * nb - c_t * vt);
      const float alpha = on * P.dt * c_t, beta = on * P.dt * (c_n - c_t);
      f0.v = f0.v + f;
      f0.w = f0.w + cross(rj, f);
      const V3 m = cross(rj, nb);
      const float rr = dot(rj, rj);
      // A += alpha (|r|^2 1 - r r^T) + beta m m^T ; H += alpha rx + beta m n^T ; M += alpha 1 + beta n n^T
      Axx += alpha * (rr - rj.x * rj.x) + beta * m.x * m.x; Axy += -alpha * rj.x * rj.y + beta * m.x * m.y; Axz += -alpha * rj.x * rj.z + beta * m.x * m.z;
      Ayy += alpha * (rr - rj.y * rj.y) + beta * m.y * m.y; Ayz += -alpha * rj.y * rj.z + beta * m.y * m.z; Azz += alpha * (rr - rj.z * rj.z) + beta * m.z * m.z;
      H.m[1] += -alpha * rj.z; H.m[2] += alpha * rj.y;
      H.m[3] += alpha * rj.z;  H.m[5] += -alpha * rj.x;
      H.m[6] += -alpha * rj.y; H.m[7] += alpha * rj.x;
      addouter(H, beta, m, nb);
      Mxx += alpha + beta * nb.x * nb.x; Mxy += beta * nb.x * nb.y; Mxz += beta * nb.x * nb.z;
      Myy += alpha + beta * nb.y * nb.y; Myz += beta * nb.y * nb.z; Mzz += alpha + beta * nb.z * nb.z;
    }
  }
  if (!hx_any(any_on != 0.f)) return false;
  // the four lanes of the side add their partial sums (afterwards they agree bitwise).  The first shape to touch a slot in
  // this substep assigns, later ones (base sub-shapes share a slot) accumulate; both decisions are wave-uniform, lanes
  // without an active point write zeros
  auto acc = [&](int field, float val) { float& d = cb.at(out, field); val = hx_qsum(val); d = accumulate ? d + val : val; };
  acc(HX_CB_F, f0.w.x); acc(HX_CB_F + 1, f0.w.y); acc(HX_CB_F + 2, f0.w.z); acc(HX_CB_F + 3, f0.v.x); acc(HX_CB_F + 4, f0.v.y); acc(HX_CB_F + 5, f0.v.z);
  acc(HX_CB_A, Axx); acc(HX_CB_A + 1, Axy); acc(HX_CB_A + 2, Axz); acc(HX_CB_A + 3, Ayy); acc(HX_CB_A + 4, Ayz); acc(HX_CB_A + 5, Azz);
  for (int i = 0; i < 9; ++i) acc(HX_CB_H + i, H.m[i]);
  acc(HX_CB_M, Mxx); acc(HX_CB_M + 1, Mxy); acc(HX_CB_M + 2, Mxz); acc(HX_CB_M + 3, Myy); acc(HX_CB_M + 4, Myz); acc(HX_CB_M + 5, Mzz);
  return true;
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

// world-frame net contact forces: base = this side's share of the base sub-shapes (the driver sums the two shares),
// shape[slot] = the side's bodies that carry collision points, in body order
template <class M> struct SideForcesT { V3 base; V3 shape[ModelInfo<M>::NSHAPE]; };

// per-side working set of one substep
template <class M> struct SideWork {
  static constexpr int NL = M::NL, NS = ModelInfo<M>::NSHAPE;
  M3 R0; SV v0, g0;                             // base rotation, base velocity (base coords), gravity field in base coords
  SV v[NL];                                      // body velocities, body coords
  float cs_c[NL], cs_s[NL];
  SV U[NL]; float Dinv[NL], uu[NL];
  SV a[NL];                                      // accelerations relative to the gravity field
  float tau[NL];
  uint32_t touched;                              // contact-buffer slots that hold contact terms this substep (wave-uniform)
};

// world position and velocity of a side's self-collision sphere centres (what the two sides exchange)
template <class M> struct SelfProbeT { V3 c[M::NSELF > 0 ? M::NSELF : 1], v[M::NSELF > 0 ? M::NSELF : 1]; };
template <class M> HXD void self_probe(const SideConst<M>& C, const ContactBuf& cb, SelfProbeT<M>& out) {
  using MI = ModelInfo<M>;
  static_for<(M::NSELF > 0 ? M::NSELF : 0)>([&](auto qc) {
    constexpr int Q = decltype(qc)::value, B = M::SELF_BODY[Q], SL = MI::slot(B);
    static_assert(SL >= 0, "a self-collision body must carry a collision shape (its state is parked in the contact buffer)");
    const M3 Rb = cb.rot_of(SL);
    const V3 rc = M::SELF_AT_SHAPE[Q] ? ld3(C.shape(B)) : mk(0.f, 0.f, 0.f);
    const V3 vw = mk(cb.at(SL, HX_CB_V), cb.at(SL, HX_CB_V + 1), cb.at(SL, HX_CB_V + 2)), vv = mk(cb.at(SL, HX_CB_V + 3), cb.at(SL, HX_CB_V + 4), cb.at(SL, HX_CB_V + 5));
    out.c[Q] = mk(cb.at(SL, HX_CB_P), cb.at(SL, HX_CB_P + 1), cb.at(SL, HX_CB_P + 2)) + mul(Rb, rc);
    out.v[Q] = mul(Rb, vv + cross(vw, rc));
  });
}
// Sphere-sphere contact of this side's self-collision bodies with the other side's: the same normal law as the ground contact
// (capped spring on the overlap, damper on the closing speed, onset inside the contact offset), no friction; the other body's
// velocity enters explicitly, this body's own linearly-implicitly.  Every lane of the side holds the same values: no lane sums.
template <class M>
HXD void self_contact(SideWork<M>& W, const DynParams& P, const SideConst<M>& C, const ContactBuf& cb, const SelfProbeT<M>& own, const SelfProbeT<M>& oth) {
  using MI = ModelInfo<M>;
  static_for<(M::NSELF > 0 ? M::NSELF : 0)>([&](auto qc) {
    constexpr int Q = decltype(qc)::value, B = M::SELF_BODY[Q], SL = MI::slot(B);
    const V3 d = own.c[Q] - oth.c[Q];
    const float dist = sqrtf(dot(d, d));
    const V3 n = (1.0f / fmaxf(dist, 1e-6f)) * d;               // from the other body to this one
    const float c_n = P.dn + P.kn * P.dt;
    const float pen = 2.0f * M::SELF_RADIUS[Q] - dist + P.roff;
    const float vn = dot(own.v[Q] - oth.v[Q], n);
    const float spring = P.kn * fmaxf(pen, 0.f);
    const float fn0 = ((P.vdep > 0.f) ? fminf(spring, c_n * P.vdep) : spring) - c_n * (vn + fmaxf(-pen, 0.f) * P.inv_dt);
    const bool act = (pen > -P.coff) && (fn0 > 0.f);
    if (!hx_any(act)) return;
    const float on = act ? 1.f : 0.f;
    const M3 Rb = cb.rot_of(SL);
    const V3 rc = M::SELF_AT_SHAPE[Q] ? ld3(C.shape(B)) : mk(0.f, 0.f, 0.f);
    const V3 nb = mulT(Rb, n), f = (on * fn0) * nb, m = cross(rc, nb), fw = cross(rc, f);
    const float beta = on * P.dt * c_n;
    const bool accumulate = (W.touched >> SL) & 1u;
    auto acc = [&](int field, float val) { float& dd = cb.at(SL, field); dd = accumulate ? dd + val : val; };
    acc(HX_CB_F, fw.x); acc(HX_CB_F + 1, fw.y); acc(HX_CB_F + 2, fw.z); acc(HX_CB_F + 3, f.x); acc(HX_CB_F + 4, f.y); acc(HX_CB_F + 5, f.z);
    acc(HX_CB_A, beta * m.x * m.x); acc(HX_CB_A + 1, beta * m.x * m.y); acc(HX_CB_A + 2, beta * m.x * m.z);
    acc(HX_CB_A + 3, beta * m.y * m.y); acc(HX_CB_A + 4, beta * m.y * m.z); acc(HX_CB_A + 5, beta * m.z * m.z);
    const float mm[3] = {m.x, m.y, m.z}, nn[3] = {nb.x, nb.y, nb.z};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) acc(HX_CB_H + 3 * i + j, beta * mm[i] * nn[j]);
    acc(HX_CB_M, beta * nb.x * nb.x); acc(HX_CB_M + 1, beta * nb.x * nb.y); acc(HX_CB_M + 2, beta * nb.x * nb.z);
    acc(HX_CB_M + 3, beta * nb.y * nb.y); acc(HX_CB_M + 4, beta * nb.y * nb.z); acc(HX_CB_M + 5, beta * nb.z * nb.z);
    W.touched |= (1u << SL);
  });
}

// ---- upward half of a substep for one side, first part: kinematics and the ground-contact phase
template <class M>
HXD void side_kin(SideWork<M>& W, const DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, const ContactBuf& cb) {
  using MI = ModelInfo<M>;
  W.R0 = quat_to_mat(S.quat);
  W.v0.w = mulT(W.R0, S.angvel); W.v0.v = mulT(W.R0, S.linvel);
  W.g0.w = mk(0.f, 0.f, 0.f); W.g0.v = P.gz * row(W.R0, 2);
  // ---- pass 1: kinematics down every chain; a body whose collision shape may touch (bounding-sphere test) leaves its
  //      state in the contact buffer and sets its entry's bit
  float gap[MI::NENT];
  static_for<M::NCH>([&](auto cc) {
    constexpr int CH = decltype(cc)::value, S0 = M::CH_START[CH], LEN = M::CH_LEN[CH];
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
      if constexpr (MI::slot(B) >= 0) {
        gap[MI::slot(B)] = shape_gap(P, C.shape(B), Rc, pc); cb.put_body(MI::slot(B), W.v[B], Rc, pc);
        // the foot touches most of the time: a second stage would only repeat what its visit does
#ifndef HX_NO_GAP2
        if constexpr (B != M::FOOT) { if (hx_any(gap[MI::slot(B)] < 0.f)) gap[MI::slot(B)] = shape_gap_points<M::NPTS[B]>(P, C.shape(B), Rc, pc); }
#endif
      }
    });
  });
  for (int k = 0; k < M::BASE_NSUB; ++k) {
    gap[MI::NSHAPE + k] = shape_gap(P, C.bsub + k * (4 + 3 * M::BASE_NP), W.R0, S.pos);
#ifndef HX_NO_GAP2
    if (hx_any(gap[MI::NSHAPE + k] < 0.f)) gap[MI::NSHAPE + k] = shape_gap_points<M::BASE_NP>(P, C.bsub + k * (4 + 3 * M::BASE_NP), W.R0, S.pos);
#endif
  }
  cb.put_body(MI::NSHAPE, W.v0, W.R0, S.pos);
  uint32_t maybe = 0u;
  for (int k = 0; k < MI::NENT; ++k) maybe |= hx_any(gap[k] < 0.f) ? (1u << k) : 0u;
#if defined(HX_STEP_PROF) && defined(__HIP_DEVICE_COMPILE__)
  // [9] shapes visited; [11] sides (of the wave's 16) whose own test asked for the visit, summed; [12] visits that a single side asked for
  {
    int sides = 0, single = 0;
    for (int k = 0; k < MI::NENT; ++k) { const int c = __popcll(__ballot(gap[k] < 0.f)) / HX_LANES_PER_SIDE; sides += c; single += (c == 1); }
    if (P.prof != nullptr && threadIdx.x == 0) { P.prof[9] += __popc(maybe); P.prof[11] += sides; P.prof[12] += single; }
  }
#endif
  HX_T(P.prof, 2);
  // ---- contact phase: one runtime loop over this lane's shapes (its shape bodies, then its share of the base)
  {
    uint32_t touched = 0u;
#pragma unroll 1
    for (int e = 0; e < MI::NENT; ++e) {
      if (!((maybe >> e) & 1u)) continue;
      const int off = C.ent[3 * e], np = C.ent[3 * e + 1], slot = C.ent[3 * e + 2];
      const bool hit = contact_shape(P, C.lds + off, np, cb, slot, slot, (touched >> slot) & 1u);
      if (hit) touched |= (1u << slot);
#if defined(HX_STEP_PROF) && defined(__HIP_DEVICE_COMPILE__)
      if (P.prof != nullptr && threadIdx.x == 0) P.prof[10] += hit ? 1 : 0;      // [10] visits that found a contact (any side)
#endif
    }
    W.touched = touched;
  }
  HX_T(P.prof, 3);
}

// ---- second part: articulated inertias leaf -> root; hands (accI, accP) = the side's contribution to the base system.
// target = PD position target per joint.
template <class M>
HXD void side_art(SideWork<M>& W, const DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, const ContactBuf& cb, const float* target, SI& accI, SV& accP) {
  using MI = ModelInfo<M>;
  // ---- pass 2: articulated inertias, leaf -> root of every chain
  accI = si0(); accP = sv0();
  static_for<M::NCH>([&](auto cc) {
    constexpr int CH = decltype(cc)::value, S0 = M::CH_START[CH], LEN = M::CH_LEN[CH];
    SI chI = si0(); SV chP = sv0();
    static_for<LEN>([&](auto ic) {
      constexpr int B = S0 + LEN - 1 - decltype(ic)::value;     // last .. first
      constexpr int K = M::AXIS[B];
      SI IA = C.inertia(B);
      SV pA = rb_bias(IA, C.h(B), C.mass(B), W.v[B]);
      if (B < S0 + LEN - 1) { siadd(IA, chI); pA = pA + chP; }
      if constexpr (MI::slot(B) >= 0) {
        constexpr int SL = MI::slot(B);
        if ((W.touched >> SL) & 1u) cb.add_to(SL, IA, pA, P.gz * mk(cb.at(SL, HX_CB_R + 6), cb.at(SL, HX_CB_R + 7), cb.at(SL, HX_CB_R + 8)));
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
  // ---- this side's share of the base shapes
  if ((W.touched >> MI::NSHAPE) & 1u) cb.add_to(MI::NSHAPE, accI, accP, W.g0.v);
  HX_T(P.prof, 4);
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
HXD void side_down(SideWork<M>& W, DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, const ContactBuf& cb, SV a0, bool want_forces, SideForcesT<M>& F) {
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
    // implicit-consistent net contact forces, world frame: f0 - B a_true with the terms the contact phase left in the buffer
    F.base = mk(0.f, 0.f, 0.f);
    if ((W.touched >> MI::NSHAPE) & 1u) { SV at = a0; at.v = at.v + W.g0.v; F.base = mul(W.R0, cb.net_force(MI::NSHAPE, at)); }
    static_for<M::NL>([&](auto ic) {
      constexpr int B = decltype(ic)::value;
      if constexpr (MI::slot(B) >= 0) {
        constexpr int SL = MI::slot(B);
        F.shape[SL] = mk(0.f, 0.f, 0.f);
        if ((W.touched >> SL) & 1u) {
          const M3 Rb = cb.rot_of(SL);
          SV at = W.a[B]; at.v = at.v + P.gz * row(Rb, 2);
          F.shape[SL] = mul(Rb, cb.net_force(SL, at));
        }
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
__device__ __forceinline__ void dyn_substep(DynStateT<M>& S, const DynParams& P, const SideConst<M>& C, const ContactBuf& cb, const float* target, float mass_scale,
                     float* tau_out, bool want_forces, SideForcesT<M>& F) {
  SideWork<M> W;
  SI accI; SV accP;
  side_kin<M>(W, S, P, C, cb);
  if constexpr (M::NSELF > 0) {
    if (P.self_on) {                     // uniform
      SelfProbeT<M> own, oth;
      self_probe<M>(C, cb, own);
      for (int q = 0; q < M::NSELF; ++q) { oth.c[q] = hx_xchg(own.c[q]); oth.v[q] = hx_xchg(own.v[q]); }
      self_contact<M>(W, P, C, cb, own, oth);
    }
  }
  side_art<M>(W, S, P, C, cb, target, accI, accP);
  for (int i = 0; i < 9; ++i) {
    accI.A.m[i] += hx_xchg(accI.A.m[i]);
    accI.H.m[i] += hx_xchg(accI.H.m[i]);
    accI.M.m[i] += hx_xchg(accI.M.m[i]);
  }
  accP.w = accP.w + hx_xchg(accP.w);
  accP.v = accP.v + hx_xchg(accP.v);
  const SV a0 = base_solve<M>(C, W.v0, mass_scale, accI, accP);
  HX_T(P.prof, 5);
  side_down<M>(W, S, P, C, cb, a0, want_forces, F);
  if (want_forces) F.base = F.base + hx_xchg(F.base);
  for (int j = 0; j < M::NL; ++j) tau_out[j] = W.tau[j];
  base_integrate<M>(S, P, W.R0, W.v0, W.g0, a0);
  HX_T(P.prof, 6);
}
#endif

// One 1 ms substep of a whole robot, both sides in sequence (host build; SL / SR share the base fields, kept equal)
template <class M>
HXD void dyn_substep_pair(DynStateT<M>& SL, DynStateT<M>& SR, const DynParams& P, const SideConst<M>& CL, const SideConst<M>& CR,
                          const float* targetL, const float* targetR, float mass_scale, float* tauL, float* tauR, bool want_forces,
                          SideForcesT<M>& FL, SideForcesT<M>& FR) {
  SideWork<M> WL, WR;
  SI aI, bI; SV aP, bP;
  float bufL[ModelInfo<M>::NSLOT * HX_CB_FIELDS], bufR[ModelInfo<M>::NSLOT * HX_CB_FIELDS];
  ContactBuf cbL, cbR; cbL.base = bufL; cbL.stride = 1; cbR.base = bufR; cbR.stride = 1;
  side_kin<M>(WL, SL, P, CL, cbL);
  side_kin<M>(WR, SR, P, CR, cbR);
  if constexpr (M::NSELF > 0) {
    if (P.self_on) {
      SelfProbeT<M> pl, pr;
      self_probe<M>(CL, cbL, pl); self_probe<M>(CR, cbR, pr);
      self_contact<M>(WL, P, CL, cbL, pl, pr);
      self_contact<M>(WR, P, CR, cbR, pr, pl);
    }
  }
  side_art<M>(WL, SL, P, CL, cbL, targetL, aI, aP);
  side_art<M>(WR, SR, P, CR, cbR, targetR, bI, bP);
  siadd(aI, bI); aP = aP + bP;
  const SV a0 = base_solve<M>(CL, WL.v0, mass_scale, aI, aP);
  side_down<M>(WL, SL, P, CL, cbL, a0, want_forces, FL);
  side_down<M>(WR, SR, P, CR, cbR, a0, want_forces, FR);
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
