// hx_env.h -- the task glue of one environment step, single source for the gfx950 kernel (hx_sim.hip) and the host
// build (oracle/host/): everything the reference computes around `gym.simulate` --
//   HectorFreeEnv.step            humanoid/envs/custom/hector_env.py:158-169  (action clip / delay / noise)
//   LeggedRobot.post_physics_step legged_robot.py:118-153 (+ callback :303-335, termination :155-160,
//                                 rewards hector_env.py:277-539 in dir() order, reset :162-214 / :256-261,
//                                 observations hector_env.py:172-254)
// written per robot on whole-robot values.  On the device all eight lanes of a robot run it redundantly (identical
// inputs after the cross-side gather) and one lane stores; on the host it runs once per robot.
#pragma once
#include "../../include/hx_sim.h"
#include "hx_dyn.h"

// ---------------------------------------------------------------- state layout (floats per env), field-major [field][env]
// (10 DoF: q 13, qd 23, act 33, last_act 43, last_last_act 53, last_dof_vel 63, last_root_vel 73, cmd 79, ... size 108)
struct SLay {
  int ROOT_POS, ROOT_QUAT, LINVEL, ANGVEL, Q, QD, ACT, LAST_ACT, LAST_LAST_ACT, LAST_DOF_VEL, LAST_ROOT_VEL, CMD, AIR, LAST_CONTACT,
      FEET_H, LAST_FEET_Z, PUSH_F, PUSH_T, FRICTION, BASE_MASS, ORIGIN, BLV, BAV, EP_RET, SIZE;
  HXD constexpr SLay(int nd)
      : ROOT_POS(0), ROOT_QUAT(3), LINVEL(7), ANGVEL(10), Q(13), QD(13 + nd), ACT(13 + 2 * nd), LAST_ACT(13 + 3 * nd),
        LAST_LAST_ACT(13 + 4 * nd), LAST_DOF_VEL(13 + 5 * nd), LAST_ROOT_VEL(13 + 6 * nd), CMD(19 + 6 * nd), AIR(23 + 6 * nd),
        LAST_CONTACT(25 + 6 * nd), FEET_H(27 + 6 * nd), LAST_FEET_Z(29 + 6 * nd), PUSH_F(31 + 6 * nd), PUSH_T(33 + 6 * nd),
        FRICTION(36 + 6 * nd), BASE_MASS(37 + 6 * nd), ORIGIN(38 + 6 * nd), BLV(41 + 6 * nd), BAV(44 + 6 * nd), EP_RET(47 + 6 * nd),
        SIZE(48 + 6 * nd) {}
};

#define HX_STAT_RING 100     /* deque(maxlen=100), on_policy_runner.py:112-113 */
struct SimPtrs {
  float* st;          // [SLay.SIZE][N]
  int* ep_len;        // [N]
  float* ep_sums;     // [HX_NUM_REWARDS][N]
  float* torques;     // [ND][N]
  float* contact;     // [(1 + ND) * 3][N]
  float* bodies;      // [52][N]
  float* obs_frame;   // [N][OBSF]   (robot-major since round 4: the stacking launch reads a robot's frame as one contiguous run)
  float* priv_frame;  // [N][PRIVF]
  float* rew;         // [N]
  unsigned char* reset;    // [N]
  unsigned char* age;      // [N] real frames in the robot's observation history (1 .. frame_stack)
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
  int t_flags;                // hx_sim_set_terrain_options (ablation switches, DynParams::tflags)
  float t_wall;               // slope_treshold * horizontal_scale for mesh_type 'trimesh', 0 for 'heightfield' (no walls)
  // pooled bounds of the grid (terrain_pool_build), [t_prows][t_pcols] each: highest node / cliff flag of the 7 x 7 nodes
  // around every second node -- what DynParams::pool / poolw window into
  const float* t_pool; const float* t_poolw;
  int t_prows, t_pcols;
  // terrain curriculum (legged_robot.py:399-419): level per env, tile column per env, platform origin per tile;
  // cur_levels == nullptr = off
  int* cur_levels;            // [N]
  const int* cur_types;       // [N]
  const float* cur_origins;   // [cur_rows][cur_cols][3]
  int cur_rows, cur_cols;
  float cur_up_dist;          // terrain.env_length / 2
  float cur_down_scale;       // max_episode_length_s * 0.5
  long long* prof;            // [16] measurement builds only (-DHX_STEP_PROF): cycle counters summed over waves; else nullptr
};

struct StepArgs {
  int mode;                  // 0: step, 1: constructor reset (reset all + first observation)
  long long step_counter;    // common_step_counter AFTER the increment of this step
  uint32_t k0, k1, rng_step;
  // single-frame observation storage (include/hx_sim.h hx_sim_step_frames): frames != 0 -> the new frames go, clipped, to the
  // consumer's slots in env-major layout together with the first valid element of the row they complete; else to the
  // [frame][env] buffers the stacking kernel reads
  int frames, obs_stack, priv_stack;
  float clip;
  hx_frame_slot fs;
};

// ---------------------------------------------------------------- counter-based RNG (Philox4x32-10)
// One out-of-line copy on the device (it is called from ~40 places of the glue); the four words come back BY VALUE, i.e. in
// registers: an output array would live on the stack, and a kernel that touches scratch memory at all pays ~7 us per launch
// on this part (tools/micro/launch_gap.hip).
struct U4 { uint32_t v[4]; };
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline))
#else
inline
#endif
U4 philox4(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = hx_mulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = hx_mulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  U4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

struct Rng {
  const float* pack;   // injected [HX_RP_SIZE][N] or nullptr
  // Device: the eight lanes of a robot share the Philox work of a run of normals -- lane l generates blocks l, l + 8, ... and
  // parks the values in `share` (LDS, >= 4 * ceil(count / 4) + 4 floats of this robot), every lane reads all of them back.
  // Host / packs: share == nullptr, nlanes == 1.
  float* share; int lane, nlanes;
  int n, env;
  uint32_t gid;        // global env id: keys the counter-based generator
  uint32_t k0, k1, step;
  HXD float uni(int field) const {
    if (pack) return pack[(size_t)field * n + env];
    const U4 o = philox4(k0, k1, gid, step, (uint32_t)field, 0u);
    return (float)(o.v[0] >> 8) * (1.0f / 16777216.0f);
  }
  // N(0,1) draws of `count` consecutive fields.  One Philox call yields the four normals of fields 4b .. 4b+3 (two
  // Box-Muller pairs), so a run costs count / 4 calls.
  HXD void nrm_run(int field0, int count, float* out) const {
    if (pack) { for (int k = 0; k < count; ++k) out[k] = pack[(size_t)(field0 + k) * n + env]; return; }
    const int b0 = field0 >> 2, b1 = (field0 + count - 1) >> 2;
    auto block = [&](int b, float* z) {
      const U4 o = philox4(k0, k1, gid, step, (uint32_t)b, 1u);
      for (int h = 0; h < 2; ++h) {
        const float u1 = 1.0f - (float)(o.v[2 * h] >> 8) * (1.0f / 16777216.0f);   // (0,1]
        const float u2 = (float)(o.v[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1)), a = 6.283185307179586f * u2;
        z[2 * h] = r * cosf(a); z[2 * h + 1] = r * sinf(a);
      }
    };
    if (share != nullptr) {
      // every lane runs the same few iterations on its own blocks (no divergence), then the values meet in LDS
      for (int b = b0 + lane; b <= b1; b += nlanes) {
        float z[4];
        block(b, z);
        for (int c = 0; c < 4; ++c) share[4 * (b - b0) + c] = z[c];
      }
      hx_lds_fence();
      for (int k = 0; k < count; ++k) out[k] = share[field0 - 4 * b0 + k];
      hx_lds_fence();
      return;
    }
    for (int b = b0; b <= b1; ++b) {
      float z[4];
      block(b, z);
      for (int c = 0; c < 4; ++c) { const int k = 4 * b + c - field0; if (k >= 0 && k < count) out[k] = z[c]; }
    }
  }
};

// ---------------------------------------------------------------- small helpers (xyzw quaternions)
HXD V3 quat_rotate_inverse(const float* q, V3 v) {
  const float qw = q[3];
  const V3 qv = mk(q[0], q[1], q[2]);
  const V3 a = (2.0f * qw * qw - 1.0f) * v;
  const V3 b = (2.0f * qw) * cross(qv, v);
  const V3 c = (2.0f * dot(qv, v)) * qv;
  return a - b + c;
}
HXD V3 quat_apply(const float* q, V3 v) {
  const V3 xyz = mk(q[0], q[1], q[2]);
  const V3 t = 2.0f * cross(xyz, v);
  return v + q[3] * t + cross(xyz, t);
}
HXD float pymod(float a, float m) { float r = fmodf(a, m); return (r < 0.f) ? r + m : r; }
HXD V3 euler_xyz_wrapped(const float* q) {
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
HXD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// compile-time sizes of a task's frames and random-pack rows (include/hx_sim.h HX_RP_* are these for 10 DoF)
// hector / hector_full: privileged frame 40 + 3 ND wide, stacked 15 deep (hector_config.py:10-16).  humanoid_ppo (XBot-L): its
// own privileged frame of 37 + 3 ND = 73 values, stacked c_frame_stack = 3 deep (humanoid_config.py:40-45)
template <class M> struct TaskDims {
  static constexpr int NL = M::NL, ND = 2 * M::NL, OBSF = 11 + 3 * ND, PRIVF = (M::XBOT ? 37 : 40) + 3 * ND, PB = 5 + 3 * ND;
  static constexpr int PRIV_STACK = M::XBOT ? 3 : HX_FRAME_STACK;
  static constexpr int RP_DELAY = 0, RP_ACT_NOISE = 1, RP_CMD_A = 1 + ND, RP_PUSH = 4 + ND, RP_RESET_Q = 9 + ND, RP_RESET_XY = 9 + 2 * ND,
                       RP_CMD_B = 11 + 2 * ND, RP_OBS_NOISE = 14 + 2 * ND, RP_LEVEL = 14 + 2 * ND + OBSF;
};

// hector_env.py:158-169 : clip, delay blend, multiplicative noise ; legged_robot.py:90-91 clip.  act: previous actions in,
// this step's actions out.
template <class M>
HXD void env_actions(const hx_sim_cfg& cfg, const Rng& rng, const float* actions_row, float* act) {
  using D = TaskDims<M>;
  const float delay = rng.uni(D::RP_DELAY) * cfg.action_delay;
  float eps[D::ND];
  rng.nrm_run(D::RP_ACT_NOISE, D::ND, eps);
  for (int j = 0; j < D::ND; ++j) {
    float x = clampf(actions_row[j], -cfg.clip_actions, cfg.clip_actions);
    x = (1.0f - delay) * x + delay * act[j];
    x = x + cfg.action_noise * eps[j] * x;
    act[j] = clampf(x, -cfg.clip_actions, cfg.clip_actions);
  }
}

// every state component must be finite and below 2^20 in magnitude.  Tested on the exponent bits with integer operations:
// the device file is built with -ffast-math (finite-math-only), under which a floating-point comparison may legally be
// folded as if NaN did not exist.
template <class M> HXD bool dyn_state_bad(const DynStateT<M>& S) {
  uint32_t emax = 0u;
  auto chk = [&](float x) { const uint32_t ex = hx_fbits(x) & 0x7f800000u; emax = ex > emax ? ex : emax; };
  chk(S.pos.x); chk(S.pos.y); chk(S.pos.z); chk(S.quat[0]); chk(S.quat[1]); chk(S.quat[2]); chk(S.quat[3]);
  chk(S.linvel.x); chk(S.linvel.y); chk(S.linvel.z); chk(S.angvel.x); chk(S.angvel.y); chk(S.angvel.z);
  for (int j = 0; j < M::NL; ++j) { chk(S.q[j]); chk(S.qd[j]); }
  return emax >= ((127u + 20u) << 23);
}

// whole-robot values the glue works on (identical on both lanes of a robot after the gather)
template <class M> struct RobotVals {
  static constexpr int ND = 2 * M::NL, NS = ModelInfo<M>::NSHAPE;
  V3 pos; float quat[4]; V3 linvel, angvel;       // base after the substeps (push / reset overwrite them)
  float qa[ND], qda[ND], torques[ND], act[ND];    // DoF order: left side 0 .. NL-1, right side NL .. ND-1
  V3 f_base; V3 side_force[2][NS];                // net contact forces, world frame
  BodyOut bo[4];                                  // L knee, L foot, R knee, R foot
  float friction, base_mass;
  int ep_len;
  bool blown;
};

// post_physics_step and everything after it for one robot; stores to HBM when `writer`.
template <class M>
HXD void env_glue(const SimPtrs& p, const hx_sim_cfg& cfg, const StepArgs& A, const int n, const int e, const bool writer, const Rng& rng, RobotVals<M>& R) {
  using D = TaskDims<M>;
  using MI = ModelInfo<M>;
  constexpr int NL = D::NL, ND = D::ND, OBSF = D::OBSF, PRIVF = D::PRIVF, PB = D::PB;
  constexpr SLay SL(ND);
  constexpr int SLOT_THIGH = M::XBOT ? -1 : MI::slot(2), SLOT_TOE = MI::slot(M::FOOT);      // XBot-L: no thigh shape; only 'base_link' terminates / is penalised
#define LD(f) (p.st[(size_t)(f) * n + e])
  // the state is stored through an index the compiler cannot tie to the one it was loaded through: otherwise the 64-bit address of
  // every field read at the start of the kernel is kept (registers, then scratch) for its store at the end (18 DoF: ~60 fields)
  int e_st = e;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(e_st));
#endif
#define ST(f, val) (p.st[(size_t)(f) * n + e_st] = (val))
  float* act = R.act; float* qa = R.qa; float* qda = R.qda; const float* torques = R.torques;
  int ep_len = R.ep_len;
  bool reset = false, time_out = false;
  float rew_total = 0.f;

  // ---- glue state
  float last_act[ND], last_last_act[ND], last_dof_vel[ND], last_root_vel[6], cmd[4];
  for (int j = 0; j < ND; ++j) { last_act[j] = LD(SL.LAST_ACT + j); last_last_act[j] = LD(SL.LAST_LAST_ACT + j); last_dof_vel[j] = LD(SL.LAST_DOF_VEL + j); }
  for (int j = 0; j < 6; ++j) last_root_vel[j] = LD(SL.LAST_ROOT_VEL + j);
  for (int j = 0; j < 4; ++j) cmd[j] = LD(SL.CMD + j);
  float air[2] = {LD(SL.AIR), LD(SL.AIR + 1)};
  float last_contact[2] = {LD(SL.LAST_CONTACT), LD(SL.LAST_CONTACT + 1)};
  float feet_h[2] = {LD(SL.FEET_H), LD(SL.FEET_H + 1)};
  float last_feet_z[2] = {LD(SL.LAST_FEET_Z), LD(SL.LAST_FEET_Z + 1)};
  float push_f[2] = {LD(SL.PUSH_F), LD(SL.PUSH_F + 1)};
  float push_t[3] = {LD(SL.PUSH_T), LD(SL.PUSH_T + 1), LD(SL.PUSH_T + 2)};
  V3 origin = mk(LD(SL.ORIGIN), LD(SL.ORIGIN + 1), LD(SL.ORIGIN + 2));
  V3 base_lin_vel = mk(LD(SL.BLV), LD(SL.BLV + 1), LD(SL.BLV + 2));
  V3 base_ang_vel = mk(LD(SL.BAV), LD(SL.BAV + 1), LD(SL.BAV + 2));
  float ep_ret = LD(SL.EP_RET);
  // per-term episode sums: loaded with the rest of the glue state (one batch of independent loads), updated in registers
  // and stored once -- a read-modify-write per reward term made every term wait for its own global load
  float ep_sum[HX_NUM_REWARDS];
  for (int r = 0; r < HX_NUM_REWARDS; ++r) ep_sum[r] = p.ep_sums[(size_t)r * n + e];

  const float TWO_PI = 6.283185307179586f;
  V3 euler, pgrav;
  if (A.mode == 0) {
    // ---- post_physics_step (legged_robot.py:127-135)
    ep_len += 1;
    base_lin_vel = quat_rotate_inverse(R.quat, R.linvel);
    base_ang_vel = quat_rotate_inverse(R.quat, R.angvel);
    pgrav = quat_rotate_inverse(R.quat, mk(0.f, 0.f, -1.f));
    euler = euler_xyz_wrapped(R.quat);
    // ---- callback (legged_robot.py:303-319)
    if (ep_len % cfg.resample_interval == 0) {
      cmd[0] = (cfg.cmd_range[0][1] - cfg.cmd_range[0][0]) * rng.uni(D::RP_CMD_A) + cfg.cmd_range[0][0];
      cmd[1] = (cfg.cmd_range[1][1] - cfg.cmd_range[1][0]) * rng.uni(D::RP_CMD_A + 1) + cfg.cmd_range[1][0];
      if (cfg.heading_command) cmd[3] = (cfg.cmd_range[3][1] - cfg.cmd_range[3][0]) * rng.uni(D::RP_CMD_A + 2) + cfg.cmd_range[3][0];
      else cmd[2] = (cfg.cmd_range[2][1] - cfg.cmd_range[2][0]) * rng.uni(D::RP_CMD_A + 2) + cfg.cmd_range[2][0];
      const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.f : 0.f;
      cmd[0] *= keep; cmd[1] *= keep;
    }
    if (cfg.heading_command) {
      const V3 fwd = quat_apply(R.quat, mk(1.f, 0.f, 0.f));
      const float heading = atan2f(fwd.y, fwd.x);
      float w = pymod(cmd[3] - heading, TWO_PI);
      if (w > 3.141592653589793f) w -= TWO_PI;
      cmd[2] = clampf(0.5f * w, -1.f, 1.f);
    }
    if (cfg.push_robots && (A.step_counter % cfg.push_interval == 0)) {
      // hector_env.py:53-68 : overwrite base velocities of every env
      push_f[0] = 2.f * cfg.max_push_vel_xy * rng.uni(D::RP_PUSH) - cfg.max_push_vel_xy;
      push_f[1] = 2.f * cfg.max_push_vel_xy * rng.uni(D::RP_PUSH + 1) - cfg.max_push_vel_xy;
      R.linvel.x = push_f[0]; R.linvel.y = push_f[1];
      for (int k = 0; k < 3; ++k) push_t[k] = 2.f * cfg.max_push_ang_vel * rng.uni(D::RP_PUSH + 2 + k) - cfg.max_push_ang_vel;
      R.angvel = mk(push_t[0], push_t[1], push_t[2]);
    }
  } else {
    pgrav = mk(0, 0, -1); euler = mk(0, 0, 0);
  }

  // contact forces per body (world): only bodies with collision points can be non-zero
  const V3 f_base = R.f_base;
  V3 f_lthigh = mk(0.f, 0.f, 0.f), f_rthigh = mk(0.f, 0.f, 0.f);
  if constexpr (SLOT_THIGH >= 0) { f_lthigh = R.side_force[0][SLOT_THIGH]; f_rthigh = R.side_force[1][SLOT_THIGH]; }
  const V3 foot_f[2] = {R.side_force[0][SLOT_TOE], R.side_force[1][SLOT_TOE]};
  const V3 foot_pos[2] = {R.bo[1].pos, R.bo[3].pos}, foot_vel[2] = {R.bo[1].linvel, R.bo[3].linvel};
  const V3 knee_pos[2] = {R.bo[0].pos, R.bo[2].pos};
  bool contact[2] = {foot_f[0].z > 5.0f, foot_f[1].z > 5.0f};
  if (writer) {
    // stored first: nothing below changes them, and the knee / foot orientations and angular velocities (28 values) and the forces
    // of the other shape bodies have no other reader -- kept to the end they were spilled around the whole glue (18 DoF)
    // diagnostic tensors (contact_forces / rigid_state views of the reference)
    {
      p.contact[(size_t)0 * n + e] = R.f_base.x; p.contact[(size_t)1 * n + e] = R.f_base.y; p.contact[(size_t)2 * n + e] = R.f_base.z;
      for (int sd = 0; sd < 2; ++sd)
        static_for<NL>([&](auto ic) {
          constexpr int B = decltype(ic)::value;
          if constexpr (MI::slot(B) >= 0) {
            const int body = 1 + sd * NL + B;
            const V3 f = R.side_force[sd][MI::slot(B)];
            p.contact[(size_t)(body * 3 + 0) * n + e] = f.x;
            p.contact[(size_t)(body * 3 + 1) * n + e] = f.y;
            p.contact[(size_t)(body * 3 + 2) * n + e] = f.z;
          }
        });
    }
    for (int b = 0; b < 4; ++b) {
      float* o = p.bodies + (size_t)(b * 13) * n + e;
      o[0] = R.bo[b].pos.x; o[(size_t)1 * n] = R.bo[b].pos.y; o[(size_t)2 * n] = R.bo[b].pos.z;
      for (int k = 0; k < 4; ++k) o[(size_t)(3 + k) * n] = R.bo[b].quat[k];
      o[(size_t)7 * n] = R.bo[b].linvel.x; o[(size_t)8 * n] = R.bo[b].linvel.y; o[(size_t)9 * n] = R.bo[b].linvel.z;
      o[(size_t)10 * n] = R.bo[b].angvel.x; o[(size_t)11 * n] = R.bo[b].angvel.y; o[(size_t)12 * n] = R.bo[b].angvel.z;
    }
  }

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
    reset = (nb > 1.0f) || (nl > 1.0f) || (nr > 1.0f) || R.blown;
    if constexpr (M::NCH > 1)        // terminate_after_contacts_on also names 'shoulder', 'twist', 'roll' (hector_w_arm_config.py:35); roll has no shape
      for (int sd = 0; sd < 2; ++sd)
        for (int b = 5; b <= 6; ++b) { const V3 f = R.side_force[sd][MI::slot(b)]; reset = reset || (sqrtf(dot(f, f)) > 1.0f); }
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
      ep_sum[id] += x;
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
      const float d[6] = {last_root_vel[0] - R.linvel.x, last_root_vel[1] - R.linvel.y, last_root_vel[2] - R.linvel.z,
                          last_root_vel[3] - R.angvel.x, last_root_vel[4] - R.angvel.y, last_root_vel[5] - R.angvel.z};
      float s2 = 0; for (int k = 0; k < 6; ++k) s2 += d[k] * d[k];
      add(HX_R_BASE_ACC, expf(-sqrtf(s2) * 3.f));
    }
    if (sc[HX_R_BASE_HEIGHT] != 0.f) {
      const float mh = (foot_pos[0].z * sm[0] + foot_pos[1].z * sm[1]) / (sm[0] + sm[1]);
      const float bh = R.pos.z - (mh - 0.05f);
      add(HX_R_BASE_HEIGHT, expf(-fabsf(bh - cfg.base_height_target) * 100.f));
    }
    if (sc[HX_R_COLLISION] != 0.f)
      add(HX_R_COLLISION, (nb > 0.1f ? 1.f : 0.f) + (nl > 0.1f ? 1.f : 0.f) + (nr > 0.1f ? 1.f : 0.f));
    if (sc[HX_R_DEFAULT_JOINT_POS] != 0.f) {
      float yr = sqrtf(dq0[0] * dq0[0] + dq0[1] * dq0[1]) + sqrtf(dq0[NL] * dq0[NL] + dq0[NL + 1] * dq0[NL + 1]);   // hip yaw / roll
      yr = clampf(yr - 0.1f, 0.f, 50.f);
      float s2 = 0; for (int j = 0; j < ND; ++j) s2 += dq0[j] * dq0[j];
      float r = expf(-yr * 100.f) - 0.01f * sqrtf(s2);
      if constexpr (M::NCH > 1) {      // hector_w_arm_env.py:371-378: shoulder yaw / pitch of both arms
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
      // hector_env.py:264-275 with compute_ref_state :90-111: ref_dof_pos is what the PREVIOUS compute_observations call left,
      // i.e. the gait phase of the episode length before this step's increment (0 right after a reset).  Zero-scaled in
      // HectorCfg, where the pose is far from the reference and the term barely depends on the phase; XBot-L (default pose = 0,
      // scale 1.6) is what pins it (fixture H).  The one step after the runner overwrites episode_length_buf
      // (on_policy_runner.py:103-106) reads the new clock here, the reference the stale pose.
      const float phase = (float)(ep_len - 1) * cfg.env_dt / cfg.cycle_time;
      const float sp = sinf(TWO_PI * phase);
      float ref[ND]; for (int j = 0; j < ND; ++j) ref[j] = 0.f;      // indices 2-4 / 7-9 whatever the DoF count (hector_w_arm_env.py:107-114)
      const float s1 = cfg.target_joint_pos_scale, s2c = 2.f * s1;
      const float l = sp > 0.f ? 0.f : sp, r_ = sp < 0.f ? 0.f : sp;
      constexpr int RR = M::XBOT ? 8 : 7;                              // humanoid_env.py:131-138: 2-4 / 8-10
      ref[2] = l * s1; ref[3] = l * s2c; ref[4] = l * s1; ref[RR] = r_ * s1; ref[RR + 1] = r_ * s2c; ref[RR + 2] = r_ * s1;
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
      const float dx = R.pos.x - origin.x, dy = R.pos.y - origin.y;
      const float dist = sqrtf(dx * dx + dy * dy);
      const bool up = dist > p.cur_up_dist;
      const bool down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * p.cur_down_scale) && !up;
      int lvl = p.cur_levels[e] + (up ? 1 : 0) - (down ? 1 : 0);
      if (lvl >= p.cur_rows) lvl = hx_imin((int)(rng.uni(D::RP_LEVEL) * (float)p.cur_rows), p.cur_rows - 1);
      else lvl = hx_imax(lvl, 0);
      const float* o = p.cur_origins + ((size_t)lvl * p.cur_cols + p.cur_types[e]) * 3;
      origin = mk(o[0], o[1], o[2]);
      if (writer) { p.cur_levels[e] = lvl; ST(SL.ORIGIN, origin.x); ST(SL.ORIGIN + 1, origin.y); ST(SL.ORIGIN + 2, origin.z); }
    }
    for (int j = 0; j < ND; ++j) {
      qa[j] = cfg.default_dof_pos[j] + (0.3f * rng.uni(D::RP_RESET_Q + j) - 0.15f);
      qda[j] = 0.f;
    }
    R.pos = mk(cfg.base_init_state[0] + origin.x, cfg.base_init_state[1] + origin.y, cfg.base_init_state[2] + origin.z);
    if (cfg.custom_origins) {
      R.pos.x += 2.f * rng.uni(D::RP_RESET_XY) - 1.f;
      R.pos.y += 2.f * rng.uni(D::RP_RESET_XY + 1) - 1.f;
    }
    for (int k = 0; k < 4; ++k) R.quat[k] = cfg.base_init_state[3 + k];
    R.linvel = mk(cfg.base_init_state[7], cfg.base_init_state[8], cfg.base_init_state[9]);
    R.angvel = mk(cfg.base_init_state[10], cfg.base_init_state[11], cfg.base_init_state[12]);
    cmd[0] = (cfg.cmd_range[0][1] - cfg.cmd_range[0][0]) * rng.uni(D::RP_CMD_B) + cfg.cmd_range[0][0];
    cmd[1] = (cfg.cmd_range[1][1] - cfg.cmd_range[1][0]) * rng.uni(D::RP_CMD_B + 1) + cfg.cmd_range[1][0];
    if (cfg.heading_command) cmd[3] = (cfg.cmd_range[3][1] - cfg.cmd_range[3][0]) * rng.uni(D::RP_CMD_B + 2) + cfg.cmd_range[3][0];
    else cmd[2] = (cfg.cmd_range[2][1] - cfg.cmd_range[2][0]) * rng.uni(D::RP_CMD_B + 2) + cfg.cmd_range[2][0];
    const float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.f : 0.f;
    cmd[0] *= keep; cmd[1] *= keep;
    for (int j = 0; j < ND; ++j) { act[j] = 0.f; last_act[j] = 0.f; last_last_act[j] = 0.f; last_dof_vel[j] = 0.f; }
    air[0] = 0.f; air[1] = 0.f;
    const int finished_len = ep_len;
    ep_len = 0;
    for (int r = 0; r < HX_NUM_REWARDS; ++r) {
      if (writer && A.mode == 0 && ep_sum[r] != 0.f) hx_atomic_add(&p.stat_sum[r], ep_sum[r]);
      ep_sum[r] = 0.f;
    }
    if (A.mode == 0 && writer) {
      // Train/mean_reward and Train/mean_episode_length of the runner (on_policy_runner.py:140-154)
      const int slot = hx_atomic_add(p.stat_cnt + 1, 1) % HX_STAT_RING;
      p.stat_ring[slot] = ep_ret;
      p.stat_ring[HX_STAT_RING + slot] = (float)finished_len;
      hx_atomic_add(p.stat_cnt, 1);
      hx_atomic_add(p.num_reset, 1);
    }
    ep_ret = 0.f;
    euler = euler_xyz_wrapped(R.quat);
    pgrav = quat_rotate_inverse(R.quat, mk(0.f, 0.f, -1.f));
  }

  // ---- compute_observations (hector_env.py:172-254) : newest 41 / 70 frame only; the 15-frame stacks are built from
  // the frames by the stack kernel (device) / the caller (host)
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
    // the privileged frame leaves in two parts (its first PB entries here, the rest below) so that the whole of it, the
    // observation frame and that frame's noise are never live together (18 DoF: 94 + 65 + 65 values)
    auto put_priv = [&](int k0, int k1) {
      if (!writer) return;
      if (A.frames) { float* d = A.fs.priv + (size_t)e * A.fs.priv_env_stride; for (int k = k0; k < k1; ++k) d[k] = fminf(fmaxf(f[k], -A.clip), A.clip); }
      else for (int k = k0; k < k1; ++k) p.priv_frame[(size_t)e * PRIVF + k] = f[k];
    };
    // obs41 = [cmd5, q10, dq10, a10, ang_vel3, euler3]
    float o[OBSF];
    for (int k = 0; k < PB; ++k) o[k] = f[k];
    put_priv(0, PB);
    o[PB] = base_ang_vel.x * cfg.obs_scale_ang_vel; o[PB + 1] = base_ang_vel.y * cfg.obs_scale_ang_vel; o[PB + 2] = base_ang_vel.z * cfg.obs_scale_ang_vel;
    o[PB + 3] = euler.x * cfg.obs_scale_quat; o[PB + 4] = euler.y * cfg.obs_scale_quat; o[PB + 5] = euler.z * cfg.obs_scale_quat;
    if (cfg.add_noise) {
      float eps[OBSF];
      rng.nrm_run(D::RP_OBS_NOISE, OBSF, eps);
      for (int k = 0; k < OBSF; ++k) {
        const float sv = cfg.noise_scale_vec[k];
        if (sv != 0.f) o[k] = o[k] + eps[k] * sv * cfg.noise_level;
      }
    }
    if (writer) {
      if (A.frames) { float* d = A.fs.obs + (size_t)e * A.fs.obs_env_stride; for (int k = 0; k < OBSF; ++k) d[k] = fminf(fmaxf(o[k], -A.clip), A.clip); }
      else for (int k = 0; k < OBSF; ++k) p.obs_frame[(size_t)e * OBSF + k] = o[k];
    }
    if constexpr (M::XBOT) {
      // humanoid_env.py:218-236: [cmd 5, q, dq, a, q - ref_dof_pos, lin vel 3, ang vel 3, euler 3, push 2 + 3, friction, mass / 30,
      // stance 2, contact 2]; ref_dof_pos of THIS call (compute_ref_state :121-143 runs first)
      float ref[ND]; for (int j = 0; j < ND; ++j) ref[j] = 0.f;
      const float s1 = cfg.target_joint_pos_scale, s2c = 2.f * s1;
      const float l = sp > 0.f ? 0.f : sp, r_ = sp < 0.f ? 0.f : sp;
      ref[2] = l * s1; ref[3] = l * s2c; ref[4] = l * s1; ref[8] = r_ * s1; ref[9] = r_ * s2c; ref[10] = r_ * s1;
      if (fabsf(sp) < 0.1f) for (int j = 0; j < ND; ++j) ref[j] = 0.f;
      for (int j = 0; j < ND; ++j) f[PB + j] = qa[j] - ref[j];
      constexpr int Q = PB + ND;
      f[Q + 0] = base_lin_vel.x * cfg.obs_scale_lin_vel; f[Q + 1] = base_lin_vel.y * cfg.obs_scale_lin_vel; f[Q + 2] = base_lin_vel.z * cfg.obs_scale_lin_vel;
      f[Q + 3] = base_ang_vel.x * cfg.obs_scale_ang_vel; f[Q + 4] = base_ang_vel.y * cfg.obs_scale_ang_vel; f[Q + 5] = base_ang_vel.z * cfg.obs_scale_ang_vel;
      f[Q + 6] = euler.x * cfg.obs_scale_quat; f[Q + 7] = euler.y * cfg.obs_scale_quat; f[Q + 8] = euler.z * cfg.obs_scale_quat;
      f[Q + 9] = push_f[0]; f[Q + 10] = push_f[1]; f[Q + 11] = push_t[0]; f[Q + 12] = push_t[1]; f[Q + 13] = push_t[2];
      f[Q + 14] = R.friction; f[Q + 15] = R.base_mass / 30.f;
      f[Q + 16] = sm[0]; f[Q + 17] = sm[1]; f[Q + 18] = contact[0] ? 1.f : 0.f; f[Q + 19] = contact[1] ? 1.f : 0.f;
    } else {
      f[PB + 0] = base_lin_vel.x * cfg.obs_scale_lin_vel; f[PB + 1] = base_lin_vel.y * cfg.obs_scale_lin_vel; f[PB + 2] = base_lin_vel.z * cfg.obs_scale_lin_vel;
      f[PB + 3] = base_ang_vel.x * cfg.obs_scale_ang_vel; f[PB + 4] = base_ang_vel.y * cfg.obs_scale_ang_vel; f[PB + 5] = base_ang_vel.z * cfg.obs_scale_ang_vel;
      f[PB + 6] = euler.x * cfg.obs_scale_quat; f[PB + 7] = euler.y * cfg.obs_scale_quat; f[PB + 8] = euler.z * cfg.obs_scale_quat;
      f[PB + 9] = foot_pos[0].x; f[PB + 10] = foot_pos[0].y; f[PB + 11] = foot_pos[0].z; f[PB + 12] = foot_pos[1].x; f[PB + 13] = foot_pos[1].y; f[PB + 14] = foot_pos[1].z;
      f[PB + 15] = foot_vel[0].x; f[PB + 16] = foot_vel[0].y; f[PB + 17] = foot_vel[0].z; f[PB + 18] = foot_vel[1].x; f[PB + 19] = foot_vel[1].y; f[PB + 20] = foot_vel[1].z;
      f[PB + 21] = R.pos.x; f[PB + 22] = R.pos.y; f[PB + 23] = R.pos.z;
      f[PB + 24] = push_f[0]; f[PB + 25] = push_f[1]; f[PB + 26] = push_t[0]; f[PB + 27] = push_t[1]; f[PB + 28] = push_t[2];
      f[PB + 29] = R.friction; f[PB + 30] = R.base_mass / 30.f;
      f[PB + 31] = sm[0]; f[PB + 32] = sm[1]; f[PB + 33] = contact[0] ? 1.f : 0.f; f[PB + 34] = contact[1] ? 1.f : 0.f;
    }
    put_priv(PB, PRIVF);
  }

  // ---- bookkeeping (legged_robot.py:146-150) and store
  if (A.mode == 0) {
    for (int j = 0; j < ND; ++j) { last_last_act[j] = last_act[j]; last_act[j] = act[j]; last_dof_vel[j] = qda[j]; }
    last_root_vel[0] = R.linvel.x; last_root_vel[1] = R.linvel.y; last_root_vel[2] = R.linvel.z;
    last_root_vel[3] = R.angvel.x; last_root_vel[4] = R.angvel.y; last_root_vel[5] = R.angvel.z;
  }
  if (!writer) return;
  ST(SL.ROOT_POS, R.pos.x); ST(SL.ROOT_POS + 1, R.pos.y); ST(SL.ROOT_POS + 2, R.pos.z);
  for (int i = 0; i < 4; ++i) ST(SL.ROOT_QUAT + i, R.quat[i]);
  ST(SL.LINVEL, R.linvel.x); ST(SL.LINVEL + 1, R.linvel.y); ST(SL.LINVEL + 2, R.linvel.z);
  ST(SL.ANGVEL, R.angvel.x); ST(SL.ANGVEL + 1, R.angvel.y); ST(SL.ANGVEL + 2, R.angvel.z);
  for (int j = 0; j < ND; ++j) {
    ST(SL.Q + j, qa[j]); ST(SL.QD + j, qda[j]); ST(SL.ACT + j, act[j]); ST(SL.LAST_ACT + j, last_act[j]);
    ST(SL.LAST_LAST_ACT + j, last_last_act[j]); ST(SL.LAST_DOF_VEL + j, last_dof_vel[j]);
    p.torques[(size_t)j * n + e] = torques[j];
  }
  for (int j = 0; j < 6; ++j) ST(SL.LAST_ROOT_VEL + j, last_root_vel[j]);
  for (int j = 0; j < 4; ++j) ST(SL.CMD + j, cmd[j]);
  ST(SL.AIR, air[0]); ST(SL.AIR + 1, air[1]); ST(SL.LAST_CONTACT, last_contact[0]); ST(SL.LAST_CONTACT + 1, last_contact[1]);
  ST(SL.FEET_H, feet_h[0]); ST(SL.FEET_H + 1, feet_h[1]); ST(SL.LAST_FEET_Z, last_feet_z[0]); ST(SL.LAST_FEET_Z + 1, last_feet_z[1]);
  ST(SL.PUSH_F, push_f[0]); ST(SL.PUSH_F + 1, push_f[1]); ST(SL.PUSH_T, push_t[0]); ST(SL.PUSH_T + 1, push_t[1]); ST(SL.PUSH_T + 2, push_t[2]);
  ST(SL.BLV, base_lin_vel.x); ST(SL.BLV + 1, base_lin_vel.y); ST(SL.BLV + 2, base_lin_vel.z);
  ST(SL.BAV, base_ang_vel.x); ST(SL.BAV + 1, base_ang_vel.y); ST(SL.BAV + 2, base_ang_vel.z);
  p.ep_len[e] = ep_len;
  for (int r = 0; r < HX_NUM_REWARDS; ++r) p.ep_sums[(size_t)r * n + e] = ep_sum[r];
  ST(SL.EP_RET, ep_ret);
  p.rew[e] = rew_total;
  p.reset[e] = reset ? 1 : 0;
  p.timeout[e] = time_out ? 1 : 0;
  {
    // frames of the robot's history that are real: a reset zeroes the whole history before the new frame is appended
    // (hector_env.py:256-261 then :246-247), so the row a reset step completes holds one frame
    const int age = reset ? 1 : hx_imin((int)p.age[e] + 1, A.obs_stack);
    p.age[e] = (unsigned char)age;
    if (A.frames) { A.fs.obs_kz[e] = (A.obs_stack - age) * OBSF; A.fs.priv_kz[e] = hx_imax(A.priv_stack - age, 0) * PRIVF; }
  }
#undef LD
#undef ST
}

// grid index of the window's node (0, 0) for a robot whose base is at (bx, by)
// (even indices: the window's pool entries are then entries of the grid's pooled maps)
HXD void patch_origin(const SimPtrs& p, float bx, float by, int& oi, int& oj) {
  const int ci = (int)floorf((bx - p.t_x0) * p.t_inv_hs + 0.5f) - HX_PATCH / 2;
  const int cj = (int)floorf((by - p.t_y0) * p.t_inv_hs + 0.5f) - HX_PATCH / 2;
  oi = hx_imin(hx_imax(ci, 0), p.t_rows - HX_PATCH) & ~1;
  oj = hx_imin(hx_imax(cj, 0), p.t_cols - HX_PATCH) & ~1;
}
// Pooled maps of a height grid h[rows][cols] (metres), built once per terrain on the host: entry (I, J) covers nodes
// 2I-2 .. 2I+4 x 2J-2 .. 2J+4 (clamped to the grid): pool = the highest of them; poolw = 1 if one of the cells whose
// lowest-index corner lies in that range is a cliff cell (its four corners span more than `wall`), else 0.  Two separable
// passes.  prows = rows / 2, pcols = cols / 2.
inline void terrain_pool_build(const float* h, int rows, int cols, float wall, float* pool, float* poolw) {
  const int prows = rows / 2, pcols = cols / 2;
  float* rmax = new float[(size_t)rows * pcols];
  float* rflag = new float[(size_t)rows * pcols];
#pragma omp parallel for schedule(static)
  for (int a = 0; a < rows; ++a) {
    const int a1 = hx_imin(a + 1, rows - 1);
    for (int J = 0; J < pcols; ++J) {
      float m = -3.0e38f, f = 0.f;
      for (int t = -2; t <= 4; ++t) {
        const int b = hx_imin(hx_imax(2 * J + t, 0), cols - 1), b1 = hx_imin(b + 1, cols - 1);
        const float h00 = h[(size_t)a * cols + b];
        m = fmaxf(m, h00);
        if (wall > 0.f) {
          const float h10 = h[(size_t)a1 * cols + b], h01 = h[(size_t)a * cols + b1], h11 = h[(size_t)a1 * cols + b1];
          if (fmaxf(fmaxf(h00, h01), fmaxf(h10, h11)) - fminf(fminf(h00, h01), fminf(h10, h11)) > wall) f = 1.f;
        }
      }
      rmax[(size_t)a * pcols + J] = m; rflag[(size_t)a * pcols + J] = f;
    }
  }
#pragma omp parallel for schedule(static)
  for (int I = 0; I < prows; ++I)
    for (int J = 0; J < pcols; ++J) {
      float m = -3.0e38f, f = 0.f;
      for (int t = -2; t <= 4; ++t) {
        const int a = hx_imin(hx_imax(2 * I + t, 0), rows - 1);
        m = fmaxf(m, rmax[(size_t)a * pcols + J]);
        f = fmaxf(f, rflag[(size_t)a * pcols + J]);
      }
      pool[(size_t)I * pcols + J] = m; poolw[(size_t)I * pcols + J] = f;
    }
  delete[] rmax; delete[] rflag;
}
HXD DynParams dyn_params(const hx_sim_cfg& cfg, float friction) {
  DynParams P;
  P.dt = cfg.sim_dt; P.inv_dt = 1.0f / cfg.sim_dt; P.gz = cfg.gravity_z;
  P.vdep = cfg.max_depenetration_velocity; P.coff = cfg.contact_offset; P.roff = cfg.rest_offset; P.tflags = 0; P.self_on = cfg.self_collisions; P.kn = cfg.contact_kn; P.dn = cfg.contact_dn; P.veps = cfg.friction_veps;
  P.lim_k = cfg.limit_k; P.lim_d = cfg.limit_d; P.mu = 0.5f * (cfg.terrain_mu + friction);
  P.patch = nullptr; P.pool = nullptr; P.poolw = nullptr; P.prof = nullptr; P.pt0 = 0; P.ptstep = 1; P.px0 = 0.f; P.py0 = 0.f; P.inv_hs = 0.f; P.wall = 0.f;
  return P;
}
