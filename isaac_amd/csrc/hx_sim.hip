// hx_sim.hip -- the hector environment step as ONE kernel launch per env step (+ one coalesced
// frame-stack kernel), and its C ABI (include/hx_sim.h).
//
// Restates, per lane (= one environment), the arithmetic of the reference's
//   HectorFreeEnv.step            humanoid/envs/custom/hector_env.py:158-169
//   LeggedRobot.step              humanoid/envs/base/legged_robot.py:84-108
//   LeggedRobot.post_physics_step legged_robot.py:118-153 (+ callback :303-335, termination :155-160,
//                                 rewards hector_env.py:277-539 in dir() order, reset :162-214 / :256-261,
//                                 observations hector_env.py:172-254)
// with the ten `gym.simulate` substeps (legged_robot.py:93-100) replaced by hx_dyn.h.
// State is SoA in HBM: field-major [field][env], so every load/store of a wave is one coalesced line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/hx_sim.h"
#include "hx_dyn.h"
#include "hx_common.h"

// ---------------------------------------------------------------- state layout (floats per env)
// (10 DoF: q 13, qd 23, act 33, last_act 43, last_last_act 53, last_dof_vel 63, last_root_vel 73, cmd 79, ... size 108)
struct SLay {
  int ROOT_POS, ROOT_QUAT, LINVEL, ANGVEL, Q, QD, ACT, LAST_ACT, LAST_LAST_ACT, LAST_DOF_VEL, LAST_ROOT_VEL, CMD, AIR, LAST_CONTACT,
      FEET_H, LAST_FEET_Z, PUSH_F, PUSH_T, FRICTION, BASE_MASS, ORIGIN, BLV, BAV, EP_RET, SIZE;
  __host__ __device__ constexpr SLay(int nd)
      : ROOT_POS(0), ROOT_QUAT(3), LINVEL(7), ANGVEL(10), Q(13), QD(13 + nd), ACT(13 + 2 * nd), LAST_ACT(13 + 3 * nd),
        LAST_LAST_ACT(13 + 4 * nd), LAST_DOF_VEL(13 + 5 * nd), LAST_ROOT_VEL(13 + 6 * nd), CMD(19 + 6 * nd), AIR(23 + 6 * nd),
        LAST_CONTACT(25 + 6 * nd), FEET_H(27 + 6 * nd), LAST_FEET_Z(29 + 6 * nd), PUSH_F(31 + 6 * nd), PUSH_T(33 + 6 * nd),
        FRICTION(36 + 6 * nd), BASE_MASS(37 + 6 * nd), ORIGIN(38 + 6 * nd), BLV(41 + 6 * nd), BAV(44 + 6 * nd), EP_RET(47 + 6 * nd),
        SIZE(48 + 6 * nd) {}
};

#define HX_STAT_RING 100     /* deque(maxlen=100), on_policy_runner.py:112-113 */
struct SimPtrs {
  float* st;          // [S_STATE_SIZE][N]
  int* ep_len;        // [N]
  float* ep_sums;     // [HX_NUM_REWARDS][N]
  float* torques;     // [10][N]
  float* contact;     // [33][N]
  float* bodies;      // [52][N]
  float* obs_frame;   // [41][N]
  float* priv_frame;  // [70][N]
  float* rew;         // [N]
  unsigned char* reset;    // [N]
  unsigned char* timeout;  // [N]
  int* num_reset;     // [1]
  // episode statistics as the runner logs them (legged_robot.py:198-201 + on_policy_runner.py:140-154,181-195):
  float* stat_sum;    // [HX_NUM_REWARDS] this step's sums of the per-term episode sums over the envs that reset
  float* stat_last;   // [HX_NUM_REWARDS] extras["episode"] of the most recent step with a reset (the dict persists in between)
  float* stat_acc;    // [HX_NUM_REWARDS] sum of stat_last over the steps since the last hx_sim_episode_stats call
  int* stat_steps;    // [2] steps accumulated ; whether stat_last has ever been set
  float* stat_ring;   // [2][HX_STAT_RING] returns / lengths of the last finished episodes (rewbuffer / lenbuffer deques)
  int* stat_cnt;      // [2] episodes finished since the last call ; ring head (total episodes ever)
  // terrain height grid (metres), row-major [t_rows][t_cols], node (i, j) at world (t_x0 + i hs, t_y0 + j hs);
  // nullptr = ground plane
  const float* terrain;
  int t_rows, t_cols;
  float t_inv_hs, t_hs, t_x0, t_y0;
  // terrain curriculum (legged_robot.py:399-419): level per env, tile column per env, platform origin per tile;
  // cur_levels == nullptr = off
  int* cur_levels;            // [N]
  const int* cur_types;       // [N]
  const float* cur_origins;   // [cur_rows][cur_cols][3]
  int cur_rows, cur_cols;
  float cur_up_dist;          // terrain.env_length / 2
  float cur_down_scale;       // max_episode_length_s * 0.5
};

// ---------------------------------------------------------------- counter-based RNG (Philox4x32-10)
__device__ __attribute__((noinline)) void philox4(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
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

struct Rng {
  const float* pack;   // injected [HX_RP_SIZE][N] or nullptr
  int n, env;
  uint32_t gid;        // global env id: keys the counter-based generator
  uint32_t k0, k1, step;
  __device__ __forceinline__ float uni(int field) const {
    if (pack) return pack[(size_t)field * n + env];
    uint32_t o[4];
    philox4(k0, k1, gid, step, (uint32_t)field, 0u, o);
    return (float)(o[0] >> 8) * (1.0f / 16777216.0f);
  }
  __device__ __forceinline__ float nrm(int field) const {
    if (pack) return pack[(size_t)field * n + env];
    uint32_t o[4];
    philox4(k0, k1, gid, step, (uint32_t)field, 1u, o);
    const float u1 = 1.0f - (float)(o[0] >> 8) * (1.0f / 16777216.0f);   // (0,1]
    const float u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
  }
};

// ---------------------------------------------------------------- small helpers (xyzw quaternions)
__device__ __forceinline__ V3 quat_rotate_inverse(const float* q, V3 v) {
  const float qw = q[3];
  const V3 qv = mk(q[0], q[1], q[2]);
  const V3 a = (2.0f * qw * qw - 1.0f) * v;
  const V3 b = (2.0f * qw) * cross(qv, v);
  const V3 c = (2.0f * dot(qv, v)) * qv;
  return a - b + c;
}
__device__ __forceinline__ V3 quat_apply(const float* q, V3 v) {
  const V3 xyz = mk(q[0], q[1], q[2]);
  const V3 t = 2.0f * cross(xyz, v);
  return v + q[3] * t + cross(xyz, t);
}
__device__ __forceinline__ float pymod(float a, float m) { float r = fmodf(a, m); return (r < 0.f) ? r + m : r; }
__device__ __forceinline__ V3 euler_xyz_wrapped(const float* q) {
  const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  const float TWO_PI = 6.283185307179586f, PI = 3.141592653589793f;
  float roll = atan2f(2.0f * (qw * qx + qy * qz), qw * qw - qx * qx - qy * qy + qz * qz);
  const float sinp = 2.0f * (qw * qy - qz * qx);
  float pitch = (fabsf(sinp) >= 1.0f) ? copysignf(1.5707963267948966f, sinp) : asinf(sinp);
  float yaw = atan2f(2.0f * (qw * qz + qx * qy), qw * qw + qx * qx - qy * qy - qz * qz);
  V3 e = mk(pymod(roll, TWO_PI), pymod(pitch, TWO_PI), pymod(yaw, TWO_PI));
  if (e.x > PI) e.x -= TWO_PI;
  if (e.y > PI) e.y -= TWO_PI;
  if (e.z > PI) e.z -= TWO_PI;
  return e;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

struct StepArgs {
  int mode;                  // 0: step, 1: constructor reset (reset all + first observation)
  long long step_counter;    // common_step_counter AFTER the increment of this step
  uint32_t k0, k1, rng_step;
};

#define LD(f) (p.st[(size_t)(f) * n + e])
#define ST(f, val) (p.st[(size_t)(f) * n + e] = (val))

template <class M>
__global__ void __launch_bounds__(64) hx_env_step_kernel(SimPtrs p, const hx_sim_cfg* __restrict__ cfgp, const float* __restrict__ actions,
                                                         const float* __restrict__ pack, StepArgs A) {
  // lane pair (2e, 2e+1) = (left leg, right leg) of robot e; 32 robots per 64-lane workgroup
  constexpr int NL = M::NL, ND = 2 * NL, OBSF = 11 + 3 * ND, PRIVF = 40 + 3 * ND, PB = 5 + 3 * ND;   // joints per lane / robot, frame widths
  constexpr SLay SL(ND);
  constexpr int S_ROOT_POS = SL.ROOT_POS, S_ROOT_QUAT = SL.ROOT_QUAT, S_LINVEL = SL.LINVEL, S_ANGVEL = SL.ANGVEL, S_Q = SL.Q, S_QD = SL.QD,
                S_ACT = SL.ACT, S_LAST_ACT = SL.LAST_ACT, S_LAST_LAST_ACT = SL.LAST_LAST_ACT, S_LAST_DOF_VEL = SL.LAST_DOF_VEL,
                S_LAST_ROOT_VEL = SL.LAST_ROOT_VEL, S_CMD = SL.CMD, S_AIR = SL.AIR, S_LAST_CONTACT = SL.LAST_CONTACT, S_FEET_H = SL.FEET_H,
                S_LAST_FEET_Z = SL.LAST_FEET_Z, S_PUSH_F = SL.PUSH_F, S_PUSH_T = SL.PUSH_T, S_FRICTION = SL.FRICTION,
                S_BASE_MASS = SL.BASE_MASS, S_ORIGIN = SL.ORIGIN, S_BLV = SL.BLV, S_BAV = SL.BAV, S_EP_RET = SL.EP_RET;
  // random-pack rows (include/hx_sim.h HX_RP_* are these for 10 DoF)
  constexpr int RP_DELAY = 0, RP_ACT_NOISE = 1, RP_CMD_A = 1 + ND, RP_PUSH = 4 + ND, RP_RESET_Q = 9 + ND, RP_RESET_XY = 9 + 2 * ND,
                RP_CMD_B = 11 + 2 * ND, RP_OBS_NOISE = 14 + 2 * ND, RP_LEVEL = 14 + 2 * ND + OBSF;
  __shared__ float lds_const[HX_LDS_CONST_FLOATS_OF(M)];
  __shared__ float lds_patch[32 * HX_PATCH * HX_PATCH];
  __shared__ int lds_patch_org[32][2];
  dyn_stage_constants<M>(lds_const, threadIdx.x, 64);
  const hx_sim_cfg& cfg = *cfgp;
  const int n = cfg.num_envs;
  const int e = (blockIdx.x * 64 + threadIdx.x) >> 1;
  const int leg = threadIdx.x & 1;
  const bool use_terrain = (p.terrain != nullptr) && (A.mode == 0);
  if (use_terrain) {
    // window of the height grid around each robot's base, fetched cooperatively: one wave instruction covers
    // four 64-byte rows of one robot's window
    if (leg == 0) {
      const int ec = min(e, n - 1);
      const float bx = p.st[(size_t)S_ROOT_POS * n + ec], by = p.st[(size_t)(S_ROOT_POS + 1) * n + ec];
      const int ci = (int)floorf((bx - p.t_x0) * p.t_inv_hs + 0.5f) - HX_PATCH / 2;
      const int cj = (int)floorf((by - p.t_y0) * p.t_inv_hs + 0.5f) - HX_PATCH / 2;
      lds_patch_org[threadIdx.x >> 1][0] = min(max(ci, 0), p.t_rows - HX_PATCH);
      lds_patch_org[threadIdx.x >> 1][1] = min(max(cj, 0), p.t_cols - HX_PATCH);
    }
    __syncthreads();
    for (int r = 0; r < 32; ++r) {
      const int oi = lds_patch_org[r][0], oj = lds_patch_org[r][1];
#pragma unroll
      for (int k = 0; k < HX_PATCH * HX_PATCH / 64; ++k) {
        const int idx = k * 64 + threadIdx.x;
        lds_patch[r * HX_PATCH * HX_PATCH + idx] = p.terrain[(size_t)(oi + idx / HX_PATCH) * p.t_cols + (oj + idx % HX_PATCH)];
      }
    }
  }
  __syncthreads();
  if (e >= n) return;                      // both lanes of a pair leave together
  const bool writer = (leg == 0);          // env-level results are computed by both lanes, stored by one
  SideConst<M> C; C.t = lds_const + leg * M::STRIDE; C.basept = lds_const + 2 * M::STRIDE;
  Rng rng; rng.pack = pack; rng.n = n; rng.env = e; rng.gid = (uint32_t)(e + cfg.env_id_offset); rng.k0 = A.k0; rng.k1 = A.k1; rng.step = A.rng_step;

  // ---- load state: base (both lanes) + this lane's leg
  DynStateT<M> S;
  S.pos = mk(LD(S_ROOT_POS), LD(S_ROOT_POS + 1), LD(S_ROOT_POS + 2));
  for (int i = 0; i < 4; ++i) S.quat[i] = LD(S_ROOT_QUAT + i);
  S.linvel = mk(LD(S_LINVEL), LD(S_LINVEL + 1), LD(S_LINVEL + 2));
  S.angvel = mk(LD(S_ANGVEL), LD(S_ANGVEL + 1), LD(S_ANGVEL + 2));
  for (int j = 0; j < NL; ++j) { S.q[j] = LD(S_Q + leg * NL + j); S.qd[j] = LD(S_QD + leg * NL + j); }
  // only what the physics needs is loaded before the substep loop; the glue state is loaded after it
  float act[ND];
  for (int j = 0; j < ND; ++j) act[j] = LD(S_ACT + j);
  const float friction = LD(S_FRICTION), base_mass = LD(S_BASE_MASS);
  int ep_len = p.ep_len[e];

  float tau_leg[NL];
  for (int j = 0; j < NL; ++j) tau_leg[j] = 0.f;
  SideForcesT<M> F; F.base = mk(0, 0, 0);
  for (int q = 0; q < M::NSHAPE; ++q) F.shape[q] = mk(0, 0, 0);
  bool reset = false, time_out = false, blown = false;
  float rew_total = 0.f;

  if (A.mode == 0) {
    // ---- hector_env.py:158-169 : clip, delay blend, multiplicative noise ; legged_robot.py:90-91 clip
    float a[ND];
    const float delay = rng.uni(RP_DELAY) * cfg.action_delay;
    for (int j = 0; j < ND; ++j) {
      float x = clampf(actions[(size_t)e * ND + j], -cfg.clip_actions, cfg.clip_actions);
      x = (1.0f - delay) * x + delay * act[j];
      x = x + cfg.action_noise * rng.nrm(RP_ACT_NOISE + j) * x;
      a[j] = clampf(x, -cfg.clip_actions, cfg.clip_actions);
    }
    for (int j = 0; j < ND; ++j) act[j] = a[j];
    // ---- legged_robot.py:93-100 : decimation x {PD torque, simulate}
    DynParams P;
    P.dt = cfg.sim_dt; P.gz = cfg.gravity_z; P.kn = cfg.contact_kn; P.dn = cfg.contact_dn; P.veps = cfg.friction_veps;
    P.lim_k = cfg.limit_k; P.lim_d = cfg.limit_d; P.mu = 0.5f * (cfg.terrain_mu + friction);
    P.patch = nullptr; P.px0 = 0.f; P.py0 = 0.f; P.inv_hs = 0.f; P.zmax = 0.f; P.zmax_near = 0.f;
    if (use_terrain) {
      const int r = threadIdx.x >> 1;
      P.patch = lds_patch + r * HX_PATCH * HX_PATCH;
      P.px0 = p.t_x0 + (float)lds_patch_org[r][0] * p.t_hs;
      P.py0 = p.t_y0 + (float)lds_patch_org[r][1] * p.t_hs;
      P.inv_hs = p.t_inv_hs;
      float zm = -3.0e38f, zn = -3.0e38f;
      for (int k = 0; k < HX_PATCH * HX_PATCH / 2; ++k) {
        const int idx = leg * (HX_PATCH * HX_PATCH / 2) + k, i = idx / HX_PATCH, j = idx % HX_PATCH;
        const float hv = P.patch[idx];
        zm = fmaxf(zm, hv);
        // nodes HX_PATCH/4 .. HX_PATCH - HX_PATCH/4 bound every point with patch coordinates in that closed range
        if (i >= HX_PATCH / 4 && i <= HX_PATCH - HX_PATCH / 4 && j >= HX_PATCH / 4 && j <= HX_PATCH - HX_PATCH / 4) zn = fmaxf(zn, hv);
      }
      P.zmax = fmaxf(zm, xchg(zm));
      P.zmax_near = fmaxf(zn, xchg(zn));
    }
    float target[NL], kpl[NL], kdl[NL], tll[NL];
    for (int j = 0; j < NL; ++j) {
      const float aj = leg ? act[NL + j] : act[j];
      target[j] = aj * cfg.action_scale + cfg.default_dof_pos[leg * NL + j];
      kpl[j] = cfg.p_gains[leg * NL + j]; kdl[j] = cfg.d_gains[leg * NL + j]; tll[j] = cfg.torque_limits[leg * NL + j];
    }
    const float mass_scale = base_mass / M::mass0();
#pragma unroll 1
    for (int sub = 0; sub < cfg.decimation; ++sub)
      dyn_substep(S, P, C, leg, target, kpl, kdl, tll, mass_scale, tau_leg, sub == cfg.decimation - 1, F);
    // Blow-up guard (no reference counterpart; PhysX clamps internally).  A non-finite or runaway state would put NaNs
    // into the observations and from there into every weight.  Such a robot is put back on its start pose with zero
    // forces right here, so nothing downstream sees the bad numbers, and the step ends its episode as a fall.
    {
      // every state component must be finite and below 2^20 in magnitude.  Tested on the exponent bits with integer
      // operations: this file is built with -ffast-math (finite-math-only), under which a floating-point comparison
      // may legally be folded as if NaN did not exist.
      uint32_t emax = 0u;
      auto chk = [&](float x) { const uint32_t ex = __float_as_uint(x) & 0x7f800000u; emax = ex > emax ? ex : emax; };
      chk(S.pos.x); chk(S.pos.y); chk(S.pos.z); chk(S.quat[0]); chk(S.quat[1]); chk(S.quat[2]); chk(S.quat[3]);
      chk(S.linvel.x); chk(S.linvel.y); chk(S.linvel.z); chk(S.angvel.x); chk(S.angvel.y); chk(S.angvel.z);
      for (int j = 0; j < NL; ++j) { chk(S.q[j]); chk(S.qd[j]); }
      float bad = (emax >= ((127u + 20u) << 23)) ? 1.f : 0.f;
      bad = fmaxf(bad, xchg(bad));
      if (bad != 0.f) {
        blown = true;
        S.pos = mk(cfg.base_init_state[0] + LD(S_ORIGIN), cfg.base_init_state[1] + LD(S_ORIGIN + 1), cfg.base_init_state[2] + LD(S_ORIGIN + 2));
        for (int k = 0; k < 4; ++k) S.quat[k] = cfg.base_init_state[3 + k];
        S.linvel = mk(0, 0, 0); S.angvel = mk(0, 0, 0);
        for (int j = 0; j < NL; ++j) { S.q[j] = cfg.default_dof_pos[leg * NL + j]; S.qd[j] = 0.f; tau_leg[j] = 0.f; }
        F.base = mk(0, 0, 0);
        for (int q = 0; q < M::NSHAPE; ++q) F.shape[q] = mk(0, 0, 0);
      }
    }
  }
  // rigid_body_state of this leg's knee / foot (post-step pose; at construction: the actor creation pose)
  BodyOut bo[4];     // L_calf, L_toe, R_calf, R_toe
  {
    BodyOut calf, toe;
    dyn_body_states(S, C, calf, toe);
    auto swap_in = [&](const BodyOut& own, BodyOut& left, BodyOut& right) {
      BodyOut oth;
      oth.pos = xchg(own.pos); oth.linvel = xchg(own.linvel); oth.angvel = xchg(own.angvel);
      for (int k = 0; k < 4; ++k) oth.quat[k] = xchg(own.quat[k]);
      left = leg ? oth : own; right = leg ? own : oth;
    };
    swap_in(calf, bo[0], bo[2]);
    swap_in(toe, bo[1], bo[3]);
  }
  // whole-robot joint vectors in DOF order (left side 0 .. NL-1, right side NL .. ND-1), identical on both lanes
  float qa[ND], qda[ND], torques[ND];
  for (int j = 0; j < NL; ++j) {
    const float oq = xchg(S.q[j]), oqd = xchg(S.qd[j]), ot = xchg(tau_leg[j]);
    qa[j] = leg ? oq : S.q[j];       qa[NL + j] = leg ? S.q[j] : oq;
    qda[j] = leg ? oqd : S.qd[j];    qda[NL + j] = leg ? S.qd[j] : oqd;
    torques[j] = leg ? ot : tau_leg[j]; torques[NL + j] = leg ? tau_leg[j] : ot;
  }
  // net contact force per collision shape: [side][slot] (slot 0 thigh, 1 toe, with arms 2 twist, 3 shoulder, 4 elbow)
  V3 side_force[2][M::NSHAPE];
  for (int q = 0; q < M::NSHAPE; ++q) {
    const V3 oth = xchg(F.shape[q]);
    side_force[0][q] = leg ? oth : F.shape[q]; side_force[1][q] = leg ? F.shape[q] : oth;
  }

  // ---- glue state
  float last_act[ND], last_last_act[ND], last_dof_vel[ND], last_root_vel[6], cmd[4];
  for (int j = 0; j < ND; ++j) { last_act[j] = LD(S_LAST_ACT + j); last_last_act[j] = LD(S_LAST_LAST_ACT + j); last_dof_vel[j] = LD(S_LAST_DOF_VEL + j); }
  for (int j = 0; j < 6; ++j) last_root_vel[j] = LD(S_LAST_ROOT_VEL + j);
  for (int j = 0; j < 4; ++j) cmd[j] = LD(S_CMD + j);
  float air[2] = {LD(S_AIR), LD(S_AIR + 1)};
  float last_contact[2] = {LD(S_LAST_CONTACT), LD(S_LAST_CONTACT + 1)};
  float feet_h[2] = {LD(S_FEET_H), LD(S_FEET_H + 1)};
  float last_feet_z[2] = {LD(S_LAST_FEET_Z), LD(S_LAST_FEET_Z + 1)};
  float push_f[2] = {LD(S_PUSH_F), LD(S_PUSH_F + 1)};
  float push_t[3] = {LD(S_PUSH_T), LD(S_PUSH_T + 1), LD(S_PUSH_T + 2)};
  V3 origin = mk(LD(S_ORIGIN), LD(S_ORIGIN + 1), LD(S_ORIGIN + 2));
  V3 base_lin_vel = mk(LD(S_BLV), LD(S_BLV + 1), LD(S_BLV + 2));
  V3 base_ang_vel = mk(LD(S_BAV), LD(S_BAV + 1), LD(S_BAV + 2));
  float ep_ret = LD(S_EP_RET);

  const float TWO_PI = 6.283185307179586f;
  V3 euler, pgrav;
  if (A.mode == 0) {
    // ---- post_physics_step (legged_robot.py:127-135)
    ep_len += 1;
    base_lin_vel = quat_rotate_inverse(S.quat, S.linvel);
    base_ang_vel = quat_rotate_inverse(S.quat, S.angvel);
    pgrav = quat_rotate_inverse(S.quat, mk(0.f, 0.f, -1.f));
    euler = euler_xyz_wrapped(S.quat);
    // ---- callback (legged_robot.py:303-319)
    if (ep_len % cfg.resample_interval == 0) {
      cmd[0] = (cfg.cmd_range[0][1] - cfg.cmd_range[0][0]) * rng.uni(RP_CMD_A) + cfg.cmd_range[0][0];
      cmd[1] = (cfg.cmd_range[1][1] - cfg.cmd_range[1][0]) * rng.uni(RP_CMD_A + 1) + cfg.cmd_range[1][0];
      if (cfg.heading_command) cmd[3] = (cfg.cmd_range[3][1] - cfg.cmd_range[3][0]) * rng.uni(RP_CMD_A + 2) + cfg.cmd_range[3][0];
      else cmd[2] = (cfg.cmd_range[2][1] - cfg.cmd_range[2][0]) * rng.uni(RP_CMD_A + 2) + cfg.cmd_range[2][0];
      const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.f : 0.f;
      cmd[0] *= keep; cmd[1] *= keep;
    }
    if (cfg.heading_command) {
      const V3 fwd = quat_apply(S.quat, mk(1.f, 0.f, 0.f));
      const float heading = atan2f(fwd.y, fwd.x);
      float w = pymod(cmd[3] - heading, TWO_PI);
      if (w > 3.141592653589793f) w -= TWO_PI;
      cmd[2] = clampf(0.5f * w, -1.f, 1.f);
    }
    if (cfg.push_robots && (A.step_counter % cfg.push_interval == 0)) {
      // hector_env.py:53-68 : overwrite base velocities of every env
      push_f[0] = 2.f * cfg.max_push_vel_xy * rng.uni(RP_PUSH) - cfg.max_push_vel_xy;
      push_f[1] = 2.f * cfg.max_push_vel_xy * rng.uni(RP_PUSH + 1) - cfg.max_push_vel_xy;
      S.linvel.x = push_f[0]; S.linvel.y = push_f[1];
      for (int k = 0; k < 3; ++k) push_t[k] = 2.f * cfg.max_push_ang_vel * rng.uni(RP_PUSH + 2 + k) - cfg.max_push_ang_vel;
      S.angvel = mk(push_t[0], push_t[1], push_t[2]);
    }
  } else {
    pgrav = mk(0, 0, -1); euler = mk(0, 0, 0);
  }

  // contact forces per body (world): only shape bodies can be non-zero
  const V3 f_base = F.base, f_lthigh = side_force[0][0], f_rthigh = side_force[1][0];
  const V3 foot_f[2] = {side_force[0][1], side_force[1][1]};
  const V3 foot_pos[2] = {bo[1].pos, bo[3].pos}, foot_vel[2] = {bo[1].linvel, bo[3].linvel};
  const V3 knee_pos[2] = {bo[0].pos, bo[2].pos};
  bool contact[2] = {foot_f[0].z > 5.0f, foot_f[1].z > 5.0f};

  auto stance_mask = [&](int len, float* sm) {
    const float phase = (float)len * cfg.env_dt / cfg.cycle_time;
    const float sp = sinf(TWO_PI * phase);
    sm[0] = (sp >= 0.f) ? 1.f : 0.f;
    sm[1] = (sp < 0.f) ? 1.f : 0.f;
    if (fabsf(sp) < 0.1f) { sm[0] = 1.f; sm[1] = 1.f; }
  };

  if (A.mode == 0) {
    // ---- termination (legged_robot.py:155-160)
    const float nb = sqrtf(dot(f_base, f_base)), nl = sqrtf(dot(f_lthigh, f_lthigh)), nr = sqrtf(dot(f_rthigh, f_rthigh));
    reset = (nb > 1.0f) || (nl > 1.0f) || (nr > 1.0f) || blown;
    if constexpr (M::ARMS)        // terminate_after_contacts_on also names 'shoulder', 'twist', 'roll' (hector_w_arm_config.py:35); roll has no shape
      for (int sd = 0; sd < 2; ++sd)
        for (int q = 2; q <= 3; ++q) reset = reset || (sqrtf(dot(side_force[sd][q], side_force[sd][q])) > 1.0f);
    time_out = (float)ep_len > cfg.max_episode_length;
    reset = reset || time_out;

    // ---- rewards, alphabetical order (legged_robot.py:216-234 ; functions hector_env.py:264-539)
    float sm[2];
    stance_mask(ep_len, sm);
    float dq0[ND];
    for (int j = 0; j < ND; ++j) dq0[j] = qa[j] - cfg.default_dof_pos[j];
    const float* sc = cfg.reward_scale;
    float rsum = 0.f;
    auto add = [&](int id, float r) {
      const float x = r * sc[id];
      rsum += x;
      if (writer) p.ep_sums[(size_t)id * n + e] += x;
    };
    if (sc[HX_R_ACTION_SMOOTHNESS] != 0.f) {
      float t1 = 0, t2 = 0, t3 = 0;
      for (int j = 0; j < ND; ++j) {
        const float d1 = last_act[j] - act[j]; t1 += d1 * d1;
        const float d2 = act[j] + last_last_act[j] - 2.f * last_act[j]; t2 += d2 * d2;
        t3 += fabsf(act[j]);
      }
      add(HX_R_ACTION_SMOOTHNESS, t1 + t2 + 0.05f * t3);
    }
    if (sc[HX_R_BASE_ACC] != 0.f) {
      const float d[6] = {last_root_vel[0] - S.linvel.x, last_root_vel[1] - S.linvel.y, last_root_vel[2] - S.linvel.z,
                          last_root_vel[3] - S.angvel.x, last_root_vel[4] - S.angvel.y, last_root_vel[5] - S.angvel.z};
      float s2 = 0; for (int k = 0; k < 6; ++k) s2 += d[k] * d[k];
      add(HX_R_BASE_ACC, expf(-sqrtf(s2) * 3.f));
    }
    if (sc[HX_R_BASE_HEIGHT] != 0.f) {
      const float mh = (foot_pos[0].z * sm[0] + foot_pos[1].z * sm[1]) / (sm[0] + sm[1]);
      const float bh = S.pos.z - (mh - 0.05f);
      add(HX_R_BASE_HEIGHT, expf(-fabsf(bh - cfg.base_height_target) * 100.f));
    }
    if (sc[HX_R_COLLISION] != 0.f)
      add(HX_R_COLLISION, (nb > 0.1f ? 1.f : 0.f) + (nl > 0.1f ? 1.f : 0.f) + (nr > 0.1f ? 1.f : 0.f));
    if (sc[HX_R_DEFAULT_JOINT_POS] != 0.f) {
      float yr = sqrtf(dq0[0] * dq0[0] + dq0[1] * dq0[1]) + sqrtf(dq0[NL] * dq0[NL] + dq0[NL + 1] * dq0[NL + 1]);   // hip yaw / roll
      yr = clampf(yr - 0.1f, 0.f, 50.f);
      float s2 = 0; for (int j = 0; j < ND; ++j) s2 += dq0[j] * dq0[j];
      float r = expf(-yr * 100.f) - 0.01f * sqrtf(s2);
      if constexpr (M::ARMS) {      // hector_w_arm_env.py:371-378: shoulder yaw / pitch of both arms
        float ar = sqrtf(dq0[5] * dq0[5] + dq0[6] * dq0[6]) + sqrtf(dq0[NL + 5] * dq0[NL + 5] + dq0[NL + 6] * dq0[NL + 6]);
        ar = clampf(ar - 0.1f, 0.f, 25.f);
        r += expf(-ar * 2.f);
      }
      add(HX_R_DEFAULT_JOINT_POS, r);
    }
    if (sc[HX_R_DOF_ACC] != 0.f) {
      float s2 = 0; for (int j = 0; j < ND; ++j) { const float d = (last_dof_vel[j] - qda[j]) / cfg.env_dt; s2 += d * d; }
      add(HX_R_DOF_ACC, s2);
    }
    if (sc[HX_R_DOF_VEL] != 0.f) {
      float s2 = 0; for (int j = 0; j < ND; ++j) s2 += qda[j] * qda[j];
      add(HX_R_DOF_VEL, s2);
    }
    if (sc[HX_R_FEET_AIR_TIME] != 0.f) {
      float r = 0;
      for (int k = 0; k < 2; ++k) {
        const bool filt = contact[k] || (sm[k] > 0.f) || (last_contact[k] != 0.f);
        last_contact[k] = contact[k] ? 1.f : 0.f;
        const float first = ((air[k] > 0.f) && filt) ? 1.f : 0.f;
        air[k] += cfg.env_dt;
        r += clampf(air[k], 0.f, 0.5f) * first;
        air[k] *= filt ? 0.f : 1.f;
      }
      add(HX_R_FEET_AIR_TIME, r);
    }
    if (sc[HX_R_FEET_CLEARANCE] != 0.f) {
      float r = 0;
      for (int k = 0; k < 2; ++k) {
        const float fz = foot_pos[k].z - 0.05f;
        feet_h[k] += fz - last_feet_z[k];
        last_feet_z[k] = fz;
        const float swing = 1.f - sm[k];
        r += ((fabsf(feet_h[k] - cfg.target_feet_height) < 0.01f) ? 1.f : 0.f) * swing;
        feet_h[k] *= contact[k] ? 0.f : 1.f;
      }
      add(HX_R_FEET_CLEARANCE, r);
    }
    if (sc[HX_R_FEET_CONTACT_FORCES] != 0.f) {
      float r = 0;
      for (int k = 0; k < 2; ++k) r += clampf(sqrtf(dot(foot_f[k], foot_f[k])) - cfg.max_contact_force, 0.f, 400.f);
      add(HX_R_FEET_CONTACT_FORCES, r);
    }
    if (sc[HX_R_FEET_CONTACT_NUMBER] != 0.f) {
      float r = 0;
      for (int k = 0; k < 2; ++k) r += ((contact[k] ? 1.f : 0.f) == sm[k]) ? 1.f : -0.3f;
      add(HX_R_FEET_CONTACT_NUMBER, r / 2.f);
    }
    auto dist_rew = [&](V3 a, V3 b, float maxd) {
      const float dx = a.x - b.x, dy = a.y - b.y;
      const float d = sqrtf(dx * dx + dy * dy);
      const float dmin = clampf(d - cfg.min_dist, -0.5f, 0.f), dmax = clampf(d - maxd, 0.f, 0.5f);
      return (expf(-fabsf(dmin) * 100.f) + expf(-fabsf(dmax) * 100.f)) / 2.f;
    };
    if (sc[HX_R_FEET_DISTANCE] != 0.f) add(HX_R_FEET_DISTANCE, dist_rew(foot_pos[0], foot_pos[1], cfg.max_dist));
    if (sc[HX_R_FOOT_SLIP] != 0.f) {
      float r = 0;
      for (int k = 0; k < 2; ++k) r += sqrtf(sqrtf(foot_vel[k].x * foot_vel[k].x + foot_vel[k].y * foot_vel[k].y)) * (contact[k] ? 1.f : 0.f);
      add(HX_R_FOOT_SLIP, r);
    }
    if (sc[HX_R_JOINT_POS] != 0.f) {
      // hector_env.py:264-275 with compute_ref_state :90-111 (reference pose uses the phase of the PREVIOUS
      // compute_observations call; this term is zero-scaled in HectorCfg)
      const float phase = (float)(ep_len) * cfg.env_dt / cfg.cycle_time;
      const float sp = sinf(TWO_PI * phase);
      float ref[ND]; for (int j = 0; j < ND; ++j) ref[j] = 0.f;      // indices 2-4 / 7-9 whatever the DoF count (hector_w_arm_env.py:107-114)
      const float s1 = cfg.target_joint_pos_scale, s2c = 2.f * s1;
      const float l = sp > 0.f ? 0.f : sp, r_ = sp < 0.f ? 0.f : sp;
      ref[2] = l * s1; ref[3] = l * s2c; ref[4] = l * s1; ref[7] = r_ * s1; ref[8] = r_ * s2c; ref[9] = r_ * s1;
      if (fabsf(sp) < 0.1f) for (int j = 0; j < ND; ++j) ref[j] = 0.f;
      float s2 = 0; for (int j = 0; j < ND; ++j) { const float d = qa[j] - ref[j]; s2 += d * d; }
      const float nn = sqrtf(s2);
      add(HX_R_JOINT_POS, expf(-2.f * nn) - 0.2f * clampf(nn, 0.f, 0.5f));
    }
    if (sc[HX_R_KNEE_DISTANCE] != 0.f) add(HX_R_KNEE_DISTANCE, dist_rew(knee_pos[0], knee_pos[1], cfg.max_dist / 2.f));
    if (sc[HX_R_LOW_SPEED] != 0.f) {
      const float as = fabsf(base_lin_vel.x), ac = fabsf(cmd[0]);
      const bool low = as < 0.5f * ac, high = as > 1.2f * ac;
      float r = 0.f;
      if (low) r = -1.f;
      if (high) r = 0.f;
      if (!(low || high)) r = 1.2f;
      const float sa = (base_lin_vel.x > 0.f) - (base_lin_vel.x < 0.f), sb = (cmd[0] > 0.f) - (cmd[0] < 0.f);
      if (sa != sb) r = -2.f;
      add(HX_R_LOW_SPEED, r * (fabsf(cmd[0]) > 0.1f ? 1.f : 0.f));
    }
    if (sc[HX_R_ORIENTATION] != 0.f) {
      const float a1 = expf(-(fabsf(euler.x) + fabsf(euler.y)) * 10.f);
      const float b1 = expf(-sqrtf(pgrav.x * pgrav.x + pgrav.y * pgrav.y) * 20.f);
      add(HX_R_ORIENTATION, (a1 + b1) / 2.f);
    }
    if (sc[HX_R_TORQUES] != 0.f) {
      float s2 = 0; for (int j = 0; j < ND; ++j) s2 += torques[j] * torques[j];
      add(HX_R_TORQUES, s2);
    }
    if (sc[HX_R_TRACK_VEL_HARD] != 0.f) {
      const float dx = cmd[0] - base_lin_vel.x, dy = cmd[1] - base_lin_vel.y;
      const float le = sqrtf(dx * dx + dy * dy), ae = fabsf(cmd[2] - base_ang_vel.z);
      add(HX_R_TRACK_VEL_HARD, (expf(-le * 10.f) + expf(-ae * 10.f)) / 2.f - 0.2f * (le + ae));
    }
    if (sc[HX_R_TRACKING_ANG_VEL] != 0.f) {
      const float d = cmd[2] - base_ang_vel.z;
      add(HX_R_TRACKING_ANG_VEL, expf(-(d * d) * cfg.tracking_sigma));
    }
    if (sc[HX_R_TRACKING_LIN_VEL] != 0.f) {
      const float dx = cmd[0] - base_lin_vel.x, dy = cmd[1] - base_lin_vel.y;
      add(HX_R_TRACKING_LIN_VEL, expf(-(dx * dx + dy * dy) * cfg.tracking_sigma));
    }
    if (sc[HX_R_VEL_MISMATCH_EXP] != 0.f) {
      const float lm = expf(-(base_lin_vel.z * base_lin_vel.z) * 10.f);
      const float am = expf(-sqrtf(base_ang_vel.x * base_ang_vel.x + base_ang_vel.y * base_ang_vel.y) * 5.f);
      add(HX_R_VEL_MISMATCH_EXP, (lm + am) / 2.f);
    }
    rew_total = cfg.only_positive_rewards ? fmaxf(rsum, 0.f) : rsum;
    ep_ret += rew_total;
  } else {
    reset = true;
  }

  // ---- reset_idx (legged_robot.py:162-214 ; hector_env.py:256-261)
  if (reset) {
    // _update_terrain_curriculum (legged_robot.py:399-419), skipped on the constructor's reset (init_done false):
    // walked more than half a tile -> harder row; less than half of the commanded distance -> easier row; past the
    // last row -> a random one.  Uses the commands of the finished episode (reset_idx resamples them afterwards).
    if (p.cur_levels != nullptr && A.mode == 0) {
      const float dx = S.pos.x - origin.x, dy = S.pos.y - origin.y;
      const float dist = sqrtf(dx * dx + dy * dy);
      const bool up = dist > p.cur_up_dist;
      const bool down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * p.cur_down_scale) && !up;
      int lvl = p.cur_levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
      if (lvl >= p.cur_rows) lvl = min((int)(rng.uni(RP_LEVEL) * (float)p.cur_rows), p.cur_rows - 1);
      else lvl = max(lvl, 0);
      const float* o = p.cur_origins + ((size_t)lvl * p.cur_cols + p.cur_types[e]) * 3;
      origin = mk(o[0], o[1], o[2]);
      if (writer) { p.cur_levels[e] = lvl; ST(S_ORIGIN, origin.x); ST(S_ORIGIN + 1, origin.y); ST(S_ORIGIN + 2, origin.z); }
    }
    for (int j = 0; j < ND; ++j) {
      qa[j] = cfg.default_dof_pos[j] + (0.3f * rng.uni(RP_RESET_Q + j) - 0.15f);
      qda[j] = 0.f;
    }
    S.pos = mk(cfg.base_init_state[0] + origin.x, cfg.base_init_state[1] + origin.y, cfg.base_init_state[2] + origin.z);
    if (cfg.custom_origins) {
      S.pos.x += 2.f * rng.uni(RP_RESET_XY) - 1.f;
      S.pos.y += 2.f * rng.uni(RP_RESET_XY + 1) - 1.f;
    }
    for (int k = 0; k < 4; ++k) S.quat[k] = cfg.base_init_state[3 + k];
    S.linvel = mk(cfg.base_init_state[7], cfg.base_init_state[8], cfg.base_init_state[9]);
    S.angvel = mk(cfg.base_init_state[10], cfg.base_init_state[11], cfg.base_init_state[12]);
    cmd[0] = (cfg.cmd_range[0][1] - cfg.cmd_range[0][0]) * rng.uni(RP_CMD_B) + cfg.cmd_range[0][0];
    cmd[1] = (cfg.cmd_range[1][1] - cfg.cmd_range[1][0]) * rng.uni(RP_CMD_B + 1) + cfg.cmd_range[1][0];
    if (cfg.heading_command) cmd[3] = (cfg.cmd_range[3][1] - cfg.cmd_range[3][0]) * rng.uni(RP_CMD_B + 2) + cfg.cmd_range[3][0];
    else cmd[2] = (cfg.cmd_range[2][1] - cfg.cmd_range[2][0]) * rng.uni(RP_CMD_B + 2) + cfg.cmd_range[2][0];
    const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.f : 0.f;
    cmd[0] *= keep; cmd[1] *= keep;
    for (int j = 0; j < ND; ++j) { act[j] = 0.f; last_act[j] = 0.f; last_last_act[j] = 0.f; last_dof_vel[j] = 0.f; }
    air[0] = 0.f; air[1] = 0.f;
    const int finished_len = ep_len;
    ep_len = 0;
    for (int r = 0; r < HX_NUM_REWARDS; ++r) {
      if (writer) {
        const float s = p.ep_sums[(size_t)r * n + e];
        if (A.mode == 0 && s != 0.f) atomicAdd(&p.stat_sum[r], s);
        p.ep_sums[(size_t)r * n + e] = 0.f;
      }
    }
    if (A.mode == 0 && writer) {
      // Train/mean_reward and Train/mean_episode_length of the runner (on_policy_runner.py:140-154)
      const int slot = atomicAdd(p.stat_cnt + 1, 1) % HX_STAT_RING;
      p.stat_ring[slot] = ep_ret;
      p.stat_ring[HX_STAT_RING + slot] = (float)finished_len;
      atomicAdd(p.stat_cnt, 1);
      atomicAdd(p.num_reset, 1);
    }
    ep_ret = 0.f;
    euler = euler_xyz_wrapped(S.quat);
    pgrav = quat_rotate_inverse(S.quat, mk(0.f, 0.f, -1.f));
  }

  // ---- compute_observations (hector_env.py:172-254) : newest 41 / 70 frame only; stacking is hx_stack_kernel
  {
    const float phase = (float)ep_len * cfg.env_dt / cfg.cycle_time;
    const float sp = sinf(TWO_PI * phase), cp = cosf(TWO_PI * phase);
    float sm[2];
    stance_mask(ep_len, sm);
    float f[PRIVF];
    f[0] = sp; f[1] = cp;
    f[2] = cmd[0] * cfg.obs_scale_lin_vel; f[3] = cmd[1] * cfg.obs_scale_lin_vel; f[4] = cmd[2] * cfg.obs_scale_ang_vel;
    for (int j = 0; j < ND; ++j) {
      f[5 + j] = (qa[j] - cfg.default_dof_pos[j]) * cfg.obs_scale_dof_pos;
      f[5 + ND + j] = qda[j] * cfg.obs_scale_dof_vel;
      f[5 + 2 * ND + j] = act[j];
    }
    // obs41 = [cmd5, q10, dq10, a10, ang_vel3, euler3]
    float o[OBSF];
    for (int k = 0; k < PB; ++k) o[k] = f[k];
    o[PB] = base_ang_vel.x * cfg.obs_scale_ang_vel; o[PB + 1] = base_ang_vel.y * cfg.obs_scale_ang_vel; o[PB + 2] = base_ang_vel.z * cfg.obs_scale_ang_vel;
    o[PB + 3] = euler.x * cfg.obs_scale_quat; o[PB + 4] = euler.y * cfg.obs_scale_quat; o[PB + 5] = euler.z * cfg.obs_scale_quat;
    if (cfg.add_noise)
      for (int k = 0; k < OBSF; ++k) {
        const float sv = cfg.noise_scale_vec[k];
        if (sv != 0.f) o[k] = o[k] + rng.nrm(RP_OBS_NOISE + k) * sv * cfg.noise_level;
      }
    if (writer) for (int k = 0; k < OBSF; ++k) p.obs_frame[(size_t)k * n + e] = o[k];
    f[PB + 0] = base_lin_vel.x * cfg.obs_scale_lin_vel; f[PB + 1] = base_lin_vel.y * cfg.obs_scale_lin_vel; f[PB + 2] = base_lin_vel.z * cfg.obs_scale_lin_vel;
    f[PB + 3] = base_ang_vel.x * cfg.obs_scale_ang_vel; f[PB + 4] = base_ang_vel.y * cfg.obs_scale_ang_vel; f[PB + 5] = base_ang_vel.z * cfg.obs_scale_ang_vel;
    f[PB + 6] = euler.x * cfg.obs_scale_quat; f[PB + 7] = euler.y * cfg.obs_scale_quat; f[PB + 8] = euler.z * cfg.obs_scale_quat;
    f[PB + 9] = foot_pos[0].x; f[PB + 10] = foot_pos[0].y; f[PB + 11] = foot_pos[0].z; f[PB + 12] = foot_pos[1].x; f[PB + 13] = foot_pos[1].y; f[PB + 14] = foot_pos[1].z;
    f[PB + 15] = foot_vel[0].x; f[PB + 16] = foot_vel[0].y; f[PB + 17] = foot_vel[0].z; f[PB + 18] = foot_vel[1].x; f[PB + 19] = foot_vel[1].y; f[PB + 20] = foot_vel[1].z;
    f[PB + 21] = S.pos.x; f[PB + 22] = S.pos.y; f[PB + 23] = S.pos.z;
    f[PB + 24] = push_f[0]; f[PB + 25] = push_f[1]; f[PB + 26] = push_t[0]; f[PB + 27] = push_t[1]; f[PB + 28] = push_t[2];
    f[PB + 29] = friction; f[PB + 30] = base_mass / 30.f;
    f[PB + 31] = sm[0]; f[PB + 32] = sm[1]; f[PB + 33] = contact[0] ? 1.f : 0.f; f[PB + 34] = contact[1] ? 1.f : 0.f;
    if (writer) for (int k = 0; k < PRIVF; ++k) p.priv_frame[(size_t)k * n + e] = f[k];
  }

  // ---- bookkeeping (legged_robot.py:146-150) and store
  if (A.mode == 0) {
    for (int j = 0; j < ND; ++j) { last_last_act[j] = last_act[j]; last_act[j] = act[j]; last_dof_vel[j] = qda[j]; }
    last_root_vel[0] = S.linvel.x; last_root_vel[1] = S.linvel.y; last_root_vel[2] = S.linvel.z;
    last_root_vel[3] = S.angvel.x; last_root_vel[4] = S.angvel.y; last_root_vel[5] = S.angvel.z;
  }
  if (!writer) return;
  ST(S_ROOT_POS, S.pos.x); ST(S_ROOT_POS + 1, S.pos.y); ST(S_ROOT_POS + 2, S.pos.z);
  for (int i = 0; i < 4; ++i) ST(S_ROOT_QUAT + i, S.quat[i]);
  ST(S_LINVEL, S.linvel.x); ST(S_LINVEL + 1, S.linvel.y); ST(S_LINVEL + 2, S.linvel.z);
  ST(S_ANGVEL, S.angvel.x); ST(S_ANGVEL + 1, S.angvel.y); ST(S_ANGVEL + 2, S.angvel.z);
  for (int j = 0; j < ND; ++j) {
    ST(S_Q + j, qa[j]); ST(S_QD + j, qda[j]); ST(S_ACT + j, act[j]); ST(S_LAST_ACT + j, last_act[j]);
    ST(S_LAST_LAST_ACT + j, last_last_act[j]); ST(S_LAST_DOF_VEL + j, last_dof_vel[j]);
    p.torques[(size_t)j * n + e] = torques[j];
  }
  for (int j = 0; j < 6; ++j) ST(S_LAST_ROOT_VEL + j, last_root_vel[j]);
  for (int j = 0; j < 4; ++j) ST(S_CMD + j, cmd[j]);
  ST(S_AIR, air[0]); ST(S_AIR + 1, air[1]); ST(S_LAST_CONTACT, last_contact[0]); ST(S_LAST_CONTACT + 1, last_contact[1]);
  ST(S_FEET_H, feet_h[0]); ST(S_FEET_H + 1, feet_h[1]); ST(S_LAST_FEET_Z, last_feet_z[0]); ST(S_LAST_FEET_Z + 1, last_feet_z[1]);
  ST(S_PUSH_F, push_f[0]); ST(S_PUSH_F + 1, push_f[1]); ST(S_PUSH_T, push_t[0]); ST(S_PUSH_T + 1, push_t[1]); ST(S_PUSH_T + 2, push_t[2]);
  ST(S_BLV, base_lin_vel.x); ST(S_BLV + 1, base_lin_vel.y); ST(S_BLV + 2, base_lin_vel.z);
  ST(S_BAV, base_ang_vel.x); ST(S_BAV + 1, base_ang_vel.y); ST(S_BAV + 2, base_ang_vel.z);
  p.ep_len[e] = ep_len;
  ST(S_EP_RET, ep_ret);
  p.rew[e] = rew_total;
  p.reset[e] = reset ? 1 : 0;
  p.timeout[e] = time_out ? 1 : 0;
  // diagnostic tensors (contact_forces / rigid_state views of the reference)
  {
    p.contact[(size_t)0 * n + e] = F.base.x; p.contact[(size_t)1 * n + e] = F.base.y; p.contact[(size_t)2 * n + e] = F.base.z;
    const int slot_body[5] = {2, 4, 5, 6, 8};       // side-local body of a shape slot: thigh, toe, twist, shoulder, elbow
    for (int sd = 0; sd < 2; ++sd)
      for (int q = 0; q < M::NSHAPE; ++q) {
        const int body = 1 + sd * NL + slot_body[q];
        p.contact[(size_t)(body * 3 + 0) * n + e] = side_force[sd][q].x;
        p.contact[(size_t)(body * 3 + 1) * n + e] = side_force[sd][q].y;
        p.contact[(size_t)(body * 3 + 2) * n + e] = side_force[sd][q].z;
      }
  }
  for (int b = 0; b < 4; ++b) {
    float* o = p.bodies + (size_t)(b * 13) * n + e;
    o[0] = bo[b].pos.x; o[(size_t)1 * n] = bo[b].pos.y; o[(size_t)2 * n] = bo[b].pos.z;
    for (int k = 0; k < 4; ++k) o[(size_t)(3 + k) * n] = bo[b].quat[k];
    o[(size_t)7 * n] = bo[b].linvel.x; o[(size_t)8 * n] = bo[b].linvel.y; o[(size_t)9 * n] = bo[b].linvel.z;
    o[(size_t)10 * n] = bo[b].angvel.x; o[(size_t)11 * n] = bo[b].angvel.y; o[(size_t)12 * n] = bo[b].angvel.z;
  }
}

// Frame stacking for BOTH observation streams (hector_env.py:246-254 + clip of legged_robot.py:104-107), one
// workgroup per env, coalesced along the rows:
//   dst[e][0:(S-1)*F] = reset ? 0 : src[e][F:S*F] ;  dst[e][(S-1)*F : S*F] = clip(frame[:,e]).
// dst may be the learner's rollout storage (zero-copy hand-over, hx_sim_step_ex); the same launch refreshes
// extras["time_outs"], hands reward / done / time-out to the learner's slot and recycles the reset counter.
struct StackArgs {
  const float* obs_src; float* obs_dst; const float* obs_frame;
  const float* priv_src; float* priv_dst; const float* priv_frame;
  const unsigned char* reset; const unsigned char* timeout; unsigned char* timeout_visible;
  const int* num_reset; int* num_reset_next;
  float* stat_sum; float* stat_last; float* stat_acc; int* stat_steps;
  const float* rew; float* rew_out; unsigned char* done_out; unsigned char* timeout_out;
  int n; float clip;
  int obs_f, obs_ld, priv_f, priv_ld;      // frame widths (41 / 70, or 65 / 94 with arms) and row strides
};
__global__ void __launch_bounds__(256) hx_stack_kernel(StackArgs a) {
  const int e = blockIdx.x;
  const bool rst = a.reset[e] != 0;
  {
    const int F = a.obs_f, ld = a.obs_ld, keep = (HX_FRAME_STACK - 1) * F;
    const float* s = a.obs_src + (size_t)e * ld;
    float* d = a.obs_dst + (size_t)e * ld;
    for (int k = threadIdx.x; k < ld; k += blockDim.x) {
      float v = 0.f;
      if (k < keep) v = rst ? 0.f : s[k + F];
      else if (k < keep + F) v = fminf(fmaxf(a.obs_frame[(size_t)(k - keep) * a.n + e], -a.clip), a.clip);
      d[k] = v;
    }
  }
  {
    const int F = a.priv_f, ld = a.priv_ld, keep = (HX_FRAME_STACK - 1) * F;
    const float* s = a.priv_src + (size_t)e * ld;
    float* d = a.priv_dst + (size_t)e * ld;
    for (int k = threadIdx.x; k < ld; k += blockDim.x) {
      float v = 0.f;
      if (k < keep) v = rst ? 0.f : s[k + F];
      else if (k < keep + F) v = fminf(fmaxf(a.priv_frame[(size_t)(k - keep) * a.n + e], -a.clip), a.clip);
      d[k] = v;
    }
  }
  if (threadIdx.x == 0) {
    // extras["time_outs"] is rebound only inside reset_idx, i.e. when at least one env reset this step
    // (legged_robot.py:172-173,208-209)
    unsigned char tv = a.timeout_visible[e];
    if (*a.num_reset > 0) { tv = a.timeout[e]; a.timeout_visible[e] = tv; }
    if (a.rew_out) { a.rew_out[e] = a.rew[e]; a.done_out[e] = rst ? 1 : 0; a.timeout_out[e] = tv; }
    if (e == 0) {
      *a.num_reset_next = 0;          // the other counter of the ping-pong pair: free until the next step
      // extras["episode"] is rebuilt only on steps with a reset and the runner appends the (possibly stale) dict every
      // step (on_policy_runner.py:141-142): per step the mean over that step's resets, then the mean over steps
      const int nr = *a.num_reset;
      if (nr > 0) {
        for (int r = 0; r < HX_NUM_REWARDS; ++r) { a.stat_last[r] = a.stat_sum[r] / (float)nr; a.stat_sum[r] = 0.f; }
        a.stat_steps[1] = 1;
      }
      if (a.stat_steps[1]) {
        for (int r = 0; r < HX_NUM_REWARDS; ++r) a.stat_acc[r] += a.stat_last[r];
        a.stat_steps[0] += 1;
      }
    }
  }
}

// ================================================================= host side
static thread_local std::string g_err;
extern "C" const char* hx_last_error(void) { return g_err.c_str(); }
void hx_set_error(const std::string& s) { g_err = s; }
extern "C" int hx_version(void) { return 100; }
extern "C" int hx_sync(void* stream) { HX_CHECK(hipStreamSynchronize((hipStream_t)stream)); return 0; }

struct hx_sim {
  int nd;                          // DoF count of the robot: 10 (hector) or 18 (hector with arms)
  int obs_f, priv_f, obs_ld, priv_ld;
  SLay L{10};                      // state layout for nd
  hx_sim_cfg cfg;
  hx_sim_cfg* cfg_d;
  hipStream_t stream;
  bool own_stream;
  SimPtrs p;
  float *obs[2], *priv[2];
  int cur;
  float *obs_cur, *priv_cur;      // where the current observation rows live (own buffer or the learner's storage)
  int* num_reset2[2]; int parity;
  unsigned char* timeout_visible;
  long long step_counter;
  uint32_t rng_step;
  uint64_t seed;
  std::vector<void*> allocs;
};

template <typename T> static int dalloc(hx_sim* s, T** ptr, size_t count) {
  HX_CHECK(hipMalloc((void**)ptr, count * sizeof(T)));
  HX_CHECK(hipMemset(*ptr, 0, count * sizeof(T)));
  s->allocs.push_back(*ptr);
  return 0;
}

static int sim_create_impl(const hx_sim_cfg* cfg, const float* friction_h, const float* base_mass_h, const float* origins_h,
                           const float* start_pos_h, uint64_t seed, void* stream, hx_sim* s);
extern "C" void hx_sim_destroy(hx_sim* s);
extern "C" int hx_sim_create(const hx_sim_cfg* cfg, const float* friction_h, const float* base_mass_h, const float* origins_h,
                             const float* start_pos_h, uint64_t seed, void* stream, hx_sim** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { hx_set_error("hx_sim_create: no HIP device (this library has no CPU path)"); return -1; }
  if (!cfg || cfg->num_envs <= 0) { hx_set_error("hx_sim_create: bad cfg"); return -2; }
  hx_sim* s = new hx_sim();
  const int rc = sim_create_impl(cfg, friction_h, base_mass_h, origins_h, start_pos_h, seed, stream, s);
  if (rc) { hx_sim_destroy(s); return rc; }       // nothing of a half-built simulator survives an error
  *out = s;
  return 0;
}

static int sim_create_impl(const hx_sim_cfg* cfg, const float* friction_h, const float* base_mass_h, const float* origins_h,
                           const float* start_pos_h, uint64_t seed, void* stream, hx_sim* s) {
  s->cfg = *cfg;
  s->nd = cfg->num_dof ? cfg->num_dof : HX_NUM_DOF;
  if (s->nd != HX_NUM_DOF && s->nd != HX_MAX_DOF) { hx_set_error("hx_sim_create: num_dof must be 10 (hector) or 18 (hector_full)"); return -2; }
  s->L = SLay(s->nd);
  s->obs_f = 11 + 3 * s->nd; s->priv_f = 40 + 3 * s->nd;
  s->obs_ld = (HX_FRAME_STACK * s->obs_f + 3) / 4 * 4; s->priv_ld = (HX_FRAME_STACK * s->priv_f + 3) / 4 * 4;
  s->seed = seed;
  s->step_counter = 0;
  s->rng_step = 0;
  s->cur = 0;
  if (stream) { s->stream = (hipStream_t)stream; s->own_stream = false; }
  else {
    if (const char* e = getenv("HX_SIM_CU_WORD")) {      // experiment hook: confine this simulator's stream to a CU subset
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = (uint32_t)strtoul(e, nullptr, 16);
      HX_CHECK(hipExtStreamCreateWithCUMask(&s->stream, 8, mask));
    } else {
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      HX_CHECK(hipStreamCreateWithPriority(&s->stream, hipStreamDefault, greatest));   // the rollout's critical path
    }
    s->own_stream = true;
  }
  const size_t n = cfg->num_envs;
  int rc = 0;
  rc |= dalloc(s, &s->p.st, (size_t)s->L.SIZE * n);
  rc |= dalloc(s, &s->p.ep_len, n);
  rc |= dalloc(s, &s->p.ep_sums, (size_t)HX_NUM_REWARDS * n);
  rc |= dalloc(s, &s->p.torques, (size_t)s->nd * n);
  rc |= dalloc(s, &s->p.contact, (size_t)(1 + s->nd) * 3 * n);
  rc |= dalloc(s, &s->p.bodies, 52 * n);
  rc |= dalloc(s, &s->p.obs_frame, (size_t)s->obs_f * n);
  rc |= dalloc(s, &s->p.priv_frame, (size_t)s->priv_f * n);
  rc |= dalloc(s, &s->p.rew, n);
  rc |= dalloc(s, &s->p.reset, n);
  rc |= dalloc(s, &s->p.timeout, n);
  rc |= dalloc(s, &s->num_reset2[0], 1); rc |= dalloc(s, &s->num_reset2[1], 1);
  s->p.num_reset = s->num_reset2[0]; s->parity = 0;
  rc |= dalloc(s, &s->p.stat_sum, HX_NUM_REWARDS); rc |= dalloc(s, &s->p.stat_last, HX_NUM_REWARDS); rc |= dalloc(s, &s->p.stat_acc, HX_NUM_REWARDS);
  rc |= dalloc(s, &s->p.stat_steps, 2); rc |= dalloc(s, &s->p.stat_ring, 2 * HX_STAT_RING);
  rc |= dalloc(s, &s->p.stat_cnt, 2);
  rc |= dalloc(s, &s->timeout_visible, n);
  for (int i = 0; i < 2; ++i) { rc |= dalloc(s, &s->obs[i], n * s->obs_ld); rc |= dalloc(s, &s->priv[i], n * s->priv_ld); }
  if (rc) return -3;
  s->obs_cur = s->obs[0]; s->priv_cur = s->priv[0];
  // initial state: actor creation pose, identity orientation, everything else zero; last_feet_z = 0.05 (hector_env.py:48)
  const SLay& SL_ = s->L;
  std::vector<float> st((size_t)SL_.SIZE * n, 0.f);
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 3; ++k) {
      st[(size_t)(SL_.ROOT_POS + k) * n + e] = start_pos_h ? start_pos_h[e * 3 + k] : 0.f;
      st[(size_t)(SL_.ORIGIN + k) * n + e] = origins_h ? origins_h[e * 3 + k] : 0.f;
    }
    st[(size_t)(SL_.ROOT_QUAT + 3) * n + e] = 1.f;
    st[(size_t)SL_.LAST_FEET_Z * n + e] = 0.05f;
    st[(size_t)(SL_.LAST_FEET_Z + 1) * n + e] = 0.05f;
    st[(size_t)SL_.FRICTION * n + e] = friction_h ? friction_h[e] : 1.f;
    st[(size_t)SL_.BASE_MASS * n + e] = base_mass_h ? base_mass_h[e] : (s->nd == HX_NUM_DOF ? HXM_MASS[0] : HXF_MASS0);
  }
  HX_CHECK(hipMemcpy(s->p.st, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
  if (dalloc(s, &s->cfg_d, 1)) return -3;
  HX_CHECK(hipMemcpy(s->cfg_d, &s->cfg, sizeof(hx_sim_cfg), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int hx_sim_set_terrain(hx_sim* s, const int16_t* heights_h, int32_t rows, int32_t cols, float horizontal_scale,
                                  float vertical_scale, float x0, float y0) {
  if (!s) { hx_set_error("hx_sim_set_terrain: null sim"); return -2; }
  if (!heights_h) { s->p.terrain = nullptr; return 0; }
  if (rows < HX_PATCH || cols < HX_PATCH || !(horizontal_scale > 0.f)) { hx_set_error("hx_sim_set_terrain: grid smaller than the contact window or bad scale"); return -2; }
  // metres in fp32, rounded from the double product exactly like the float32 mesh vertices of the reference
  std::vector<float> h((size_t)rows * cols);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((double)heights_h[i] * (double)vertical_scale);
  float* d = nullptr;
  HX_CHECK(hipMalloc((void**)&d, h.size() * sizeof(float)));
  s->allocs.push_back(d);
  HX_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  s->p.terrain = d; s->p.t_rows = rows; s->p.t_cols = cols;
  s->p.t_hs = horizontal_scale; s->p.t_inv_hs = 1.0f / horizontal_scale; s->p.t_x0 = x0; s->p.t_y0 = y0;
  return 0;
}

extern "C" int hx_sim_set_terrain_curriculum(hx_sim* s, const float* origins_h, int32_t rows, int32_t cols, const int32_t* levels_h,
                                             const int32_t* types_h, float env_length, float max_episode_length_s) {
  if (!s) { hx_set_error("hx_sim_set_terrain_curriculum: null sim"); return -2; }
  if (!origins_h) { s->p.cur_levels = nullptr; return 0; }
  if (rows <= 0 || cols <= 0 || !levels_h || !types_h) { hx_set_error("hx_sim_set_terrain_curriculum: bad table"); return -2; }
  const int n = s->cfg.num_envs;
  for (int e = 0; e < n; ++e)
    if (levels_h[e] < 0 || levels_h[e] >= rows || types_h[e] < 0 || types_h[e] >= cols) {
      hx_set_error("hx_sim_set_terrain_curriculum: level / type outside the tile table"); return -2;
    }
  int *lv = nullptr, *ty = nullptr; float* og = nullptr;
  if (dalloc(s, &lv, (size_t)n) || dalloc(s, &ty, (size_t)n) || dalloc(s, &og, (size_t)rows * cols * 3)) return -3;
  HX_CHECK(hipMemcpy(lv, levels_h, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  HX_CHECK(hipMemcpy(ty, types_h, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  HX_CHECK(hipMemcpy(og, origins_h, (size_t)rows * cols * 3 * sizeof(float), hipMemcpyHostToDevice));
  s->p.cur_levels = lv; s->p.cur_types = ty; s->p.cur_origins = og; s->p.cur_rows = rows; s->p.cur_cols = cols;
  s->p.cur_up_dist = 0.5f * env_length; s->p.cur_down_scale = 0.5f * max_episode_length_s;
  return 0;
}

extern "C" int hx_sim_get_terrain_levels(hx_sim* s, int32_t* levels_h) {
  if (!s || !s->p.cur_levels) { hx_set_error("hx_sim_get_terrain_levels: no terrain curriculum set"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(levels_h, s->p.cur_levels, (size_t)s->cfg.num_envs * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void hx_sim_destroy(hx_sim* s) {
  if (!s) return;
  (void)hipDeviceSynchronize();
  for (void* a : s->allocs) hipFree(a);
  if (s->own_stream) hipStreamDestroy(s->stream);
  delete s;
}

struct StepOut { float* obs; float* priv; float* rew; unsigned char* done; unsigned char* timeout; };

static int launch_step(hx_sim* s, const float* actions, const float* pack, int mode, const StepOut* out) {
  const int n = s->cfg.num_envs;
  StepArgs A;
  A.mode = mode;
  if (mode == 0) s->step_counter += 1;
  A.step_counter = s->step_counter;
  A.k0 = (uint32_t)(s->seed & 0xffffffffu);
  A.k1 = (uint32_t)(s->seed >> 32);
  A.rng_step = s->rng_step++;
  // reset counter: ping-pong pair; the stack kernel of step t zeroes the counter step t+1 will use
  s->p.num_reset = s->num_reset2[s->parity];
  if (s->nd == HX_NUM_DOF) hipLaunchKernelGGL(hx_env_step_kernel<ModelHector>, dim3((2 * n + 63) / 64), dim3(64), 0, s->stream, s->p, s->cfg_d, actions, pack, A);
  else hipLaunchKernelGGL(hx_env_step_kernel<ModelFull>, dim3((2 * n + 63) / 64), dim3(64), 0, s->stream, s->p, s->cfg_d, actions, pack, A);
  // destination of the new observation rows: the caller's (learner storage) or the other internal buffer
  float* od = s->obs[s->cur ^ 1]; float* pd = s->priv[s->cur ^ 1];
  if (s->obs_cur == od) { od = s->obs[s->cur]; pd = s->priv[s->cur]; }
  if (out && out->obs) { od = out->obs; pd = out->priv; } else s->cur ^= 1;
  StackArgs k{};
  k.obs_src = s->obs_cur; k.obs_dst = od; k.obs_frame = s->p.obs_frame;
  k.priv_src = s->priv_cur; k.priv_dst = pd; k.priv_frame = s->p.priv_frame;
  k.reset = s->p.reset; k.timeout = s->p.timeout; k.timeout_visible = s->timeout_visible;
  k.num_reset = s->num_reset2[s->parity]; k.num_reset_next = s->num_reset2[s->parity ^ 1];
  k.stat_sum = s->p.stat_sum; k.stat_last = s->p.stat_last; k.stat_acc = s->p.stat_acc; k.stat_steps = s->p.stat_steps;
  k.rew = s->p.rew; k.rew_out = out ? out->rew : nullptr; k.done_out = out ? out->done : nullptr; k.timeout_out = out ? out->timeout : nullptr;
  k.n = n; k.clip = s->cfg.clip_observations;
  k.obs_f = s->obs_f; k.obs_ld = s->obs_ld; k.priv_f = s->priv_f; k.priv_ld = s->priv_ld;
  hipLaunchKernelGGL(hx_stack_kernel, dim3(n), dim3(256), 0, s->stream, k);
  s->obs_cur = od; s->priv_cur = pd;
  s->parity ^= 1;
  HX_CHECK(hipGetLastError());
  return 0;
}

extern "C" int hx_sim_reset_all(hx_sim* s, const float* pack) { return launch_step(s, nullptr, pack, 1, nullptr); }
extern "C" int hx_sim_step(hx_sim* s, const float* actions, const float* pack) {
  if (!actions) { hx_set_error("hx_sim_step: actions is NULL"); return -2; }
  return launch_step(s, actions, pack, 0, nullptr);
}
// Zero-copy form: the new observation rows and the step's reward / done / time-out flags are written straight into
// buffers of the caller (the learner's rollout storage); any of the three scalar outputs may be NULL together.
extern "C" int hx_sim_step_ex(hx_sim* s, const float* actions, const float* pack, float* obs_dst, float* priv_dst,
                              float* rew_dst, uint8_t* done_dst, uint8_t* timeout_dst) {
  if (!actions) { hx_set_error("hx_sim_step_ex: actions is NULL"); return -2; }
  StepOut o{obs_dst, priv_dst, rew_dst, done_dst, timeout_dst};
  return launch_step(s, actions, pack, 0, &o);
}

extern "C" int hx_sim_buffer(hx_sim* s, int which, void** dptr) {
  if (!s || !dptr) { hx_set_error("hx_sim_buffer: null argument"); return -2; }
  switch (which) {
    case HX_BUF_OBS: *dptr = s->obs_cur; break;
    case HX_BUF_PRIV: *dptr = s->priv_cur; break;
    case HX_BUF_REW: *dptr = s->p.rew; break;
    case HX_BUF_RESET: *dptr = s->p.reset; break;
    case HX_BUF_TIMEOUT: *dptr = s->p.timeout; break;
    case HX_BUF_TIMEOUT_VISIBLE: *dptr = s->timeout_visible; break;
    case HX_BUF_EP_LEN: *dptr = s->p.ep_len; break;
    case HX_BUF_COMMANDS: *dptr = s->p.st + (size_t)s->L.CMD * s->cfg.num_envs; break;
    case HX_BUF_TORQUES: *dptr = s->p.torques; break;
    case HX_BUF_CONTACT: *dptr = s->p.contact; break;
    case HX_BUF_BODY_STATE: *dptr = s->p.bodies; break;
    case HX_BUF_EPISODE_SUMS: *dptr = s->p.ep_sums; break;
    case HX_BUF_FEET_AIR_TIME: *dptr = s->p.st + (size_t)s->L.AIR * s->cfg.num_envs; break;
    case HX_BUF_FEET_HEIGHT: *dptr = s->p.st + (size_t)s->L.FEET_H * s->cfg.num_envs; break;
    case HX_BUF_NUM_RESET: *dptr = s->num_reset2[s->parity ^ 1]; break;
    default: hx_set_error("hx_sim_buffer: unknown id"); return -2;
  }
  return 0;
}

extern "C" int hx_sim_get_state(hx_sim* s, float* root_h, float* q_h, float* qd_h) {
  if (!s || !root_h || !q_h || !qd_h) { hx_set_error("hx_sim_get_state: null argument"); return -2; }
  const size_t n = s->cfg.num_envs;
  const int nd = s->nd;
  std::vector<float> st((size_t)(13 + 2 * nd) * n);
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(st.data(), s->p.st, st.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 13; ++k) root_h[e * 13 + k] = st[(size_t)k * n + e];
    for (int j = 0; j < nd; ++j) { q_h[e * nd + j] = st[(size_t)(s->L.Q + j) * n + e]; qd_h[e * nd + j] = st[(size_t)(s->L.QD + j) * n + e]; }
  }
  return 0;
}

extern "C" int hx_sim_set_state(hx_sim* s, const float* root_h, const float* q_h, const float* qd_h) {
  if (!s || !root_h || !q_h || !qd_h) { hx_set_error("hx_sim_set_state: null argument"); return -2; }
  const size_t n = s->cfg.num_envs;
  const int nd = s->nd;
  std::vector<float> st((size_t)(13 + 2 * nd) * n);
  HX_CHECK(hipStreamSynchronize(s->stream));
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 13; ++k) st[(size_t)k * n + e] = root_h[e * 13 + k];
    for (int j = 0; j < nd; ++j) { st[(size_t)(s->L.Q + j) * n + e] = q_h[e * nd + j]; st[(size_t)(s->L.QD + j) * n + e] = qd_h[e * nd + j]; }
  }
  HX_CHECK(hipMemcpy(s->p.st, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int hx_sim_set_commands(hx_sim* s, const float* cmd_h) {
  if (!s || !cmd_h) { hx_set_error("hx_sim_set_commands: null argument"); return -2; }
  const size_t n = s->cfg.num_envs;
  std::vector<float> c(4 * n);
  for (size_t e = 0; e < n; ++e)
    for (int k = 0; k < 4; ++k) c[(size_t)k * n + e] = cmd_h[e * 4 + k];
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(s->p.st + (size_t)s->L.CMD * n, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int hx_sim_get_base_velocities(hx_sim* s, float* lin_h, float* ang_h) {
  if (!s || !lin_h || !ang_h) { hx_set_error("hx_sim_get_base_velocities: null argument"); return -2; }
  const size_t n = s->cfg.num_envs;
  std::vector<float> v(6 * n);
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(v.data(), s->p.st + (size_t)s->L.BLV * n, v.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < n; ++e)
    for (int k = 0; k < 3; ++k) { lin_h[e * 3 + k] = v[(size_t)k * n + e]; ang_h[e * 3 + k] = v[(size_t)(3 + k) * n + e]; }
  return 0;
}

extern "C" int hx_sim_set_episode_length(hx_sim* s, const int32_t* h) {
  if (!s || !h) { hx_set_error("hx_sim_set_episode_length: null argument"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(s->p.ep_len, h, (size_t)s->cfg.num_envs * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}
extern "C" int hx_sim_set_step_counter(hx_sim* s, int64_t c) { s->step_counter = c; return 0; }

extern "C" int hx_sim_episode_stats(hx_sim* s, float* mean_h, int32_t* count_h) {
  if (!s || !mean_h || !count_h) { hx_set_error("hx_sim_episode_stats: null argument"); return -2; }
  float acc[HX_NUM_REWARDS], ring[2 * HX_STAT_RING]; int steps[2] = {0, 0}, cnt[2] = {0, 0};
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(acc, s->p.stat_acc, sizeof(acc), hipMemcpyDeviceToHost));
  HX_CHECK(hipMemcpy(ring, s->p.stat_ring, sizeof(ring), hipMemcpyDeviceToHost));
  HX_CHECK(hipMemcpy(steps, s->p.stat_steps, sizeof(steps), hipMemcpyDeviceToHost));
  HX_CHECK(hipMemcpy(cnt, s->p.stat_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
  for (int r = 0; r < HX_NUM_REWARDS; ++r) mean_h[r] = steps[0] > 0 ? acc[r] / (float)steps[0] / s->cfg.max_episode_length_s : 0.f;
  const int filled = cnt[1] < HX_STAT_RING ? cnt[1] : HX_STAT_RING;
  double sr = 0, sl = 0;
  for (int i = 0; i < filled; ++i) { sr += ring[i]; sl += ring[HX_STAT_RING + i]; }
  mean_h[HX_NUM_REWARDS] = filled > 0 ? (float)(sr / filled) : 0.f;
  mean_h[HX_NUM_REWARDS + 1] = filled > 0 ? (float)(sl / filled) : 0.f;
  *count_h = cnt[0];
  // ep_infos.clear() of the runner (on_policy_runner.py:170); the deques and the stale extras["episode"] persist
  HX_CHECK(hipMemset(s->p.stat_acc, 0, sizeof(acc)));
  HX_CHECK(hipMemset(s->p.stat_steps, 0, sizeof(int)));
  HX_CHECK(hipMemset(s->p.stat_cnt, 0, sizeof(int)));
  return 0;
}
extern "C" void* hx_sim_stream(hx_sim* s) { return (void*)s->stream; }
