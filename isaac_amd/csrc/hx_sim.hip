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
#include "../../include/hx_lab.h"
#include "hx_env.h"
#include "hx_common.h"

// One launch per env step; the ten 1 ms substeps run inside with the robot's state in registers.  The arithmetic lives
// in hx_dyn.h (dynamics) and hx_env.h (task glue), both shared with the host build; this kernel is the lane-group driver:
// eight lanes per robot (four per body side, hx_math.h), HX_RPW = 8 robots per 64-lane workgroup.
#define HX_RPW (64 / HX_LANES_PER_ROBOT)      /* robots per wave */
#ifndef HX_ENV_WPB
#define HX_ENV_WPB 1                          /* waves per workgroup (each wave is self-contained: own LDS region, own robots) */
#endif
#define HX_PROF_WAVES 8192
template <class M> __host__ __device__ static constexpr size_t env_step_lds_bytes() {
  return sizeof(float) * ((ModelInfo<M>::LDS_FLOATS + 3) / 4 * 4 + HX_RPW * HX_PATCH_LD + 2 * HX_RPW * HX_POOL_LD + 2 * HX_RPW + (size_t)ModelInfo<M>::NSLOT * HX_CB_FIELDS * (64 / HX_LANES_PER_SIDE) + HX_RPW * 80);
}
template <class M>
__global__ void __launch_bounds__(64 * HX_ENV_WPB) hx_env_step_kernel(SimPtrs p, const hx_sim_cfg* __restrict__ cfgp, const float* __restrict__ actions,
                                                         const float* __restrict__ pack, StepArgs A) {
  using D = TaskDims<M>;
  using MI = ModelInfo<M>;
  constexpr int NL = D::NL, ND = D::ND;
  constexpr SLay SL(ND);
  // dynamic LDS: staged constants | HX_RPW height windows | pooled bounds | cliff flags | window origins | per-lane contact buffer
#if HX_ENV_WPB > 1
  extern __shared__ float lds_all0[];
  const int tidx = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* lds_all = lds_all0 + (size_t)wv * (env_step_lds_bytes<M>() / sizeof(float));
#else
  extern __shared__ float lds_all[];
  const int tidx = threadIdx.x, wv = 0;
#endif
  float* lds_const = lds_all;
  float* lds_patch = lds_const + (MI::LDS_FLOATS + 3) / 4 * 4;
  float* lds_pool = lds_patch + HX_RPW * HX_PATCH_LD;
  float* lds_poolw = lds_pool + HX_RPW * HX_POOL_LD;
  int (*lds_patch_org)[2] = reinterpret_cast<int (*)[2]>(lds_poolw + HX_RPW * HX_POOL_LD);
  float* lds_cb = lds_poolw + HX_RPW * HX_POOL_LD + 2 * HX_RPW;
#if defined(HX_STEP_PROF)
  __shared__ long long lds_prof[18];
  if (tidx < 18) lds_prof[tidx] = 0;
  __syncthreads();
  // [15]: shader clock at the start (phase timers); [16], [17]: shader-clock and 100 MHz wall-clock stamps of the whole wave -> the
  // clock the kernel ran at (MI355X_MICROARCH.md "DVFS give-back" item 6: d s_memtime / d s_memrealtime x 100 MHz)
  if (tidx == 0) { lds_prof[15] = clock64(); lds_prof[16] = -(long long)clock64(); lds_prof[17] = -(long long)wall_clock64(); }
  long long* const prof = (p.prof != nullptr) ? lds_prof : nullptr;
#else
  long long* const prof = nullptr;
#endif
  const hx_sim_cfg& cfg = *cfgp;
  dyn_stage_constants<M>(lds_const, tidx, 64, cfg.p_gains, cfg.d_gains, cfg.torque_limits, cfg.default_dof_pos);
  const int n = cfg.num_envs;
  const int e = ((int)blockIdx.x * HX_ENV_WPB + wv) * HX_RPW + (tidx >> 3);
  const int side = (tidx >> 2) & 1, sub = tidx & 3, r = tidx >> 3;      // body side, lane of the side, robot of the wave
  const bool use_terrain = (p.terrain != nullptr) && (A.mode == 0);
  if (use_terrain) {
    // Every lane fetches an eighth of its own robot's windows: 2 of the 16 rows of the height window, 1 of the 8 rows of
    // the two pooled maps.  All addresses follow from the robot's own base position, so the loads of a lane are
    // independent and stay in flight together.
    const int ec = min(e, n - 1), part = tidx & 7;
    int oi, oj;
    patch_origin(p, p.st[(size_t)SL.ROOT_POS * n + ec], p.st[(size_t)(SL.ROOT_POS + 1) * n + ec], oi, oj);
    if (part == 0) { lds_patch_org[r][0] = oi; lds_patch_org[r][1] = oj; }
    const float* src = p.terrain + (size_t)(oi + 2 * part) * p.t_cols + oj;
    float* dst = lds_patch + r * HX_PATCH_LD + 2 * part * HX_PATCH;
    {
      float v[2 * HX_PATCH];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < HX_PATCH; ++b) v[a * HX_PATCH + b] = src[(size_t)a * p.t_cols + b];
#pragma unroll
      for (int k = 0; k < 2 * HX_PATCH; ++k) dst[k] = v[k];
    }
    const size_t po = (size_t)(oi / 2 + part) * p.t_pcols + oj / 2;
#pragma unroll
    for (int b = 0; b < HX_POOL; ++b) {
      lds_pool[r * HX_POOL_LD + part * HX_POOL + b] = p.t_pool[po + b];
      lds_poolw[r * HX_POOL_LD + part * HX_POOL + b] = p.t_poolw[po + b];
    }
  }
  __syncthreads();
  HX_T(prof, 0);
  if (e >= n) return;                      // both lanes of a pair leave together
  const bool writer = (side == 0) && (sub == 0);         // env-level results are computed by all eight lanes, stored by one
  SideConst<M> C; C.bind(lds_const, side);
  Rng rng; rng.pack = pack; rng.share = lds_cb + (size_t)MI::NSLOT * HX_CB_FIELDS * (64 / HX_LANES_PER_SIDE) + r * 80; rng.lane = tidx & 7; rng.nlanes = HX_LANES_PER_ROBOT; rng.n = n; rng.env = e; rng.gid = (uint32_t)(e + cfg.env_id_offset); rng.k0 = A.k0; rng.k1 = A.k1; rng.step = A.rng_step;
#define LD(f) (p.st[(size_t)(f) * n + e])
  // ---- load state: base (both lanes) + this lane's side; only what the physics needs (the glue loads its own state)
  DynStateT<M> S;
  S.pos = mk(LD(SL.ROOT_POS), LD(SL.ROOT_POS + 1), LD(SL.ROOT_POS + 2));
  for (int i = 0; i < 4; ++i) S.quat[i] = LD(SL.ROOT_QUAT + i);
  S.linvel = mk(LD(SL.LINVEL), LD(SL.LINVEL + 1), LD(SL.LINVEL + 2));
  S.angvel = mk(LD(SL.ANGVEL), LD(SL.ANGVEL + 1), LD(SL.ANGVEL + 2));
  for (int j = 0; j < NL; ++j) { S.q[j] = LD(SL.Q + side * NL + j); S.qd[j] = LD(SL.QD + side * NL + j); }
  RobotVals<M> R;
  for (int j = 0; j < ND; ++j) R.act[j] = LD(SL.ACT + j);
  R.friction = LD(SL.FRICTION); R.base_mass = LD(SL.BASE_MASS);
  R.ep_len = p.ep_len[e];
  R.blown = false;

  float tau_side[NL];
  for (int j = 0; j < NL; ++j) tau_side[j] = 0.f;
  SideForcesT<M> F; F.base = mk(0, 0, 0);
  for (int q = 0; q < MI::NSHAPE; ++q) F.shape[q] = mk(0, 0, 0);

  if (A.mode == 0) {
    env_actions<M>(cfg, rng, actions + (size_t)e * ND, R.act);
    // ---- legged_robot.py:93-100 : decimation x {PD torque, simulate}
    DynParams P = dyn_params(cfg, R.friction);
    P.prof = prof;
    HX_T(prof, 1);
    P.pt0 = sub; P.ptstep = HX_LANES_PER_SIDE;
    if (use_terrain) {
      P.patch = lds_patch + r * HX_PATCH_LD;
      P.px0 = p.t_x0 + (float)lds_patch_org[r][0] * p.t_hs;
      P.py0 = p.t_y0 + (float)lds_patch_org[r][1] * p.t_hs;
      P.inv_hs = p.t_inv_hs; P.wall = p.t_wall; P.tflags = p.t_flags;
      P.pool = lds_pool + r * HX_POOL_LD; P.poolw = lds_poolw + r * HX_POOL_LD;
    }
    // one buffer column per body side: the four lanes of a side hold the same body states and (after the quad sums) write the
    // same contact terms, so they share it -- 16 columns per wave instead of 64 keeps the kernel's LDS footprint small enough
    // for the deferred critic's GEMM workgroups to stay resident beside it
    ContactBuf cb; cb.base = lds_cb + (tidx >> 2); cb.stride = 64 / HX_LANES_PER_SIDE;
    float target[NL];
    for (int j = 0; j < NL; ++j) {
      const float aj = side ? R.act[NL + j] : R.act[j];
      target[j] = aj * cfg.action_scale + C.q0(j);
    }
    const float mass_scale = R.base_mass / M::MASS0;
    const int decimation = cfg.decimation;
    // the processed actions are needed again by the glue: over the substeps they wait in the robot's random-number exchange area
    // (idle here) instead of in registers (18 DoF: 18 of them)
    float* const park = const_cast<float*>(rng.share);
    for (int j = 0; j < ND; ++j) park[j] = R.act[j];
#pragma unroll 1
    for (int sub = 0; sub < decimation; ++sub)
      dyn_substep<M>(S, P, C, cb, target, mass_scale, tau_side, sub == decimation - 1, F);
    for (int j = 0; j < ND; ++j) R.act[j] = park[j];
    // Blow-up guard (no reference counterpart; PhysX clamps internally).  A non-finite or runaway state would put NaNs
    // into the observations and from there into every weight.  Such a robot is put back on its start pose with zero
    // forces right here, so nothing downstream sees the bad numbers, and the step ends its episode as a fall.
    {
      float bad = dyn_state_bad<M>(S) ? 1.f : 0.f;
      bad = fmaxf(bad, hx_xchg(bad));
      if (bad != 0.f) {
        R.blown = true;
        S.pos = mk(cfg.base_init_state[0] + LD(SL.ORIGIN), cfg.base_init_state[1] + LD(SL.ORIGIN + 1), cfg.base_init_state[2] + LD(SL.ORIGIN + 2));
        for (int k = 0; k < 4; ++k) S.quat[k] = cfg.base_init_state[3 + k];
        S.linvel = mk(0, 0, 0); S.angvel = mk(0, 0, 0);
        for (int j = 0; j < NL; ++j) { S.q[j] = C.q0(j); S.qd[j] = 0.f; tau_side[j] = 0.f; }
        F.base = mk(0, 0, 0);
        for (int q = 0; q < MI::NSHAPE; ++q) F.shape[q] = mk(0, 0, 0);
      }
    }
  }
#undef LD
  // ---- gather the partner lane's side: rigid_body_state of the knees / feet (post-step pose; at construction: the actor
  //      creation pose), whole-robot joint vectors in DoF order, net contact force per shape body
  {
    BodyOut knee, foot;
    dyn_body_states<M>(S, C, knee, foot);
    auto swap_in = [&](const BodyOut& own, BodyOut& left, BodyOut& right) {
      BodyOut oth;
      oth.pos = hx_xchg(own.pos); oth.linvel = hx_xchg(own.linvel); oth.angvel = hx_xchg(own.angvel);
      for (int k = 0; k < 4; ++k) oth.quat[k] = hx_xchg(own.quat[k]);
      left = side ? oth : own; right = side ? own : oth;
    };
    swap_in(knee, R.bo[0], R.bo[2]);
    swap_in(foot, R.bo[1], R.bo[3]);
  }
  for (int j = 0; j < NL; ++j) {
    const float oq = hx_xchg(S.q[j]), oqd = hx_xchg(S.qd[j]), ot = hx_xchg(tau_side[j]);
    R.qa[j] = side ? oq : S.q[j];          R.qa[NL + j] = side ? S.q[j] : oq;
    R.qda[j] = side ? oqd : S.qd[j];       R.qda[NL + j] = side ? S.qd[j] : oqd;
    R.torques[j] = side ? ot : tau_side[j]; R.torques[NL + j] = side ? tau_side[j] : ot;
  }
  for (int q = 0; q < MI::NSHAPE; ++q) {
    const V3 oth = hx_xchg(F.shape[q]);
    R.side_force[0][q] = side ? oth : F.shape[q]; R.side_force[1][q] = side ? F.shape[q] : oth;
  }
  R.f_base = F.base;
  R.pos = S.pos; for (int k = 0; k < 4; ++k) R.quat[k] = S.quat[k];
  R.linvel = S.linvel; R.angvel = S.angvel;
  HX_T(prof, 7);
  env_glue<M>(p, cfg, A, n, e, writer, rng, R);
  HX_T(prof, 8);
#if defined(HX_STEP_PROF)
  if (prof != nullptr && tidx == 0) {
    lds_prof[16] += (long long)clock64(); lds_prof[17] += (long long)wall_clock64();
    for (int k = 0; k < 18; ++k) if (k != 15) atomicAdd((unsigned long long*)&p.prof[k], (unsigned long long)lds_prof[k]);
    if (blockIdx.x < HX_PROF_WAVES) {
      p.prof[20 + blockIdx.x] += lds_prof[17];        // per-wave lifetime (100 MHz ticks), summed over launches
      // of the LAST launch: start and end stamps (the 100 MHz counter is common to the whole device) and where the wave ran
      const long long t_end = (long long)wall_clock64();
      p.prof[20 + HX_PROF_WAVES + blockIdx.x] = t_end - lds_prof[17];
      p.prof[20 + 2 * HX_PROF_WAVES + blockIdx.x] = t_end;
      p.prof[20 + 3 * HX_PROF_WAVES + blockIdx.x] = ((long long)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);
      for (int k = 0; k < 15; ++k) p.prof[20 + (4 + k) * HX_PROF_WAVES + blockIdx.x] = lds_prof[k];     // this wave's phase cycles and contact-loop visits in the last launch
    }
  }
#endif
}


// Frame stacking for BOTH observation streams (hector_env.py:246-254 + clip of legged_robot.py:104-107), one
// workgroup per env, coalesced along the rows:
//   dst[e][0:(S-1)*F] = reset ? 0 : src[e][F:S*F] ;  dst[e][(S-1)*F : S*F] = clip(frame[e][:]).
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
  int* pause;                              // hx_sim_set_pause_word: raised by this launch, lowered by its consumer's next launch
  int obs_f, obs_ld, priv_f, priv_ld;      // frame widths (41 / 70, 65 / 94 with arms, 47 / 73 for XBot-L) and row strides
  int priv_stack;                          // frames in a privileged row: 15, or c_frame_stack = 3 for XBot-L
};
// one row:  d[0 : keep) = rst ? 0 : s[F : F + keep),  d[keep : keep + F) = clip(frame[e][:]),  d[keep + F : ld) = 0.
// Rows start 16-byte aligned and ld is a multiple of 4: every store is 16 bytes wide; the shifted source is misaligned by F % 4
// floats, so a store's four values come from two aligned 16-byte loads (the second one hits the line the first one fetched).
// Only the few 16-byte groups that touch the new frame or the padding are assembled element by element.
__device__ __forceinline__ void stack_row(const float* __restrict__ s, float* __restrict__ d, const float* __restrict__ frame, const int e, const int n,
                                          const int F, const int ld, const int keep, const bool rst, const float clip, const int tid, const int nthreads) {
  const int m = F & 3, q0 = F >> 2, nq = ld >> 2;
  const float4* s4 = reinterpret_cast<const float4*>(s);
  float4* d4 = reinterpret_cast<float4*>(d);
  for (int q = tid; q < nq; q += nthreads) {
    const int k0 = 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k0 + 3 < keep) {
      if (!rst) {
        const float4 a = s4[q + q0];
        const float4 b = s4[min(q + q0 + 1, nq - 1)];
        switch (m) {                           // uniform
          case 0: v = a; break;
          case 1: v = make_float4(a.y, a.z, a.w, b.x); break;
          case 2: v = make_float4(a.z, a.w, b.x, b.y); break;
          default: v = make_float4(a.w, b.x, b.y, b.z); break;
        }
      }
    } else {
      // clamped addresses, no branch around the loads: the eight of them are in flight together
      float x[4], fs[4], ff[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        fs[c] = s[min(k + F, ld - 1)];
        ff[c] = frame[(size_t)e * F + min(max(k - keep, 0), F - 1)];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        x[c] = (k < keep) ? (rst ? 0.f : fs[c]) : (k < keep + F) ? fminf(fmaxf(ff[c], -clip), clip) : 0.f;
      }
      v = make_float4(x[0], x[1], x[2], x[3]);
    }
    d4[q] = v;
  }
}
__global__ void __launch_bounds__(256) hx_stack_kernel(StackArgs a) {
  const int e = blockIdx.x;
  // a consumer's background work (the learner's critic on its half of the chip) sleeps while word 0 is up; word 1 tells the consumer's
  // next launch that this one raised it
  if (a.pause != nullptr && e == 0 && threadIdx.x == 0) { atomicAdd(a.pause, 1); a.pause[1] = 1; }
  const bool rst = a.reset[e] != 0;
  stack_row(a.obs_src + (size_t)e * a.obs_ld, a.obs_dst + (size_t)e * a.obs_ld, a.obs_frame, e, a.n, a.obs_f, a.obs_ld, (HX_FRAME_STACK - 1) * a.obs_f,
            rst, a.clip, (int)threadIdx.x, (int)blockDim.x);
  stack_row(a.priv_src + (size_t)e * a.priv_ld, a.priv_dst + (size_t)e * a.priv_ld, a.priv_frame, e, a.n, a.priv_f, a.priv_ld, (a.priv_stack - 1) * a.priv_f,
            rst, a.clip, (int)threadIdx.x, (int)blockDim.x);
  if (threadIdx.x == 0) {
    // extras["time_outs"] is rebound only inside reset_idx, i.e. when at least one env reset this step
    // (legged_robot.py:172-173,208-209)
    unsigned char tv = a.timeout_visible[e];
    if (*a.num_reset > 0) { tv = a.timeout[e]; a.timeout_visible[e] = tv; }
    if (a.rew_out) { a.rew_out[e] = a.rew[e]; a.done_out[e] = rst ? 1 : 0; a.timeout_out[e] = tv; }
  }
  if (e == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {
    // the other counter of the ping-pong pair is free until the next step; extras["episode"] is rebuilt only on steps with a
    // reset and the runner appends the (possibly stale) dict every step (on_policy_runner.py:141-142): per step the mean over
    // that step's resets, then the mean over steps -- one wave, one reward term per lane (hx_common.h)
    const hx_step_book b{a.reset, a.timeout, a.timeout_visible, a.num_reset, a.num_reset_next, a.stat_sum, a.stat_last, a.stat_acc, a.stat_steps,
                         a.rew, a.rew_out, a.done_out, a.timeout_out, a.n};
    hx_step_book_global(b, (int)threadIdx.x - 64);
  }
}

// ================================================================= host side
static thread_local std::string g_err;
extern "C" const char* hx_last_error(void) { return g_err.c_str(); }
void hx_set_error(const std::string& s) { g_err = s; }
// ---- validated experiment knobs (hx_common.h)
extern char** environ;
int hx_knobs_check(void) {
  // library knobs, then the names the Python host / tools / tests of this repository read themselves
  static const char* known[] = {"HX_CRITIC_CHUNK", "HX_BG_PERSIST", "HX_BG_TILE", "HX_BG_WAVES", "HX_STACK_PAUSE", "HX_CRITIC_LATE", "HX_CRITIC_CU_WORD", "HX_ACTOR_WAVES", "HX_ACTOR_ROWS", "HX_ACTOR_DEPTH", "HX_FWD_IN_TILE",
                                "HX_UPDATE_STREAMS", "HX_WGRAD_BLOCKS", "HX_WGRAD_GROUP", "HX_WGRAD_MULTI", "HX_GEMM_SP", "HX_FRAMES_GATHER", "HX_GEMM_PAIR", "HX_HEAD_MFMA", "HX_CRITIC_YIELD", "HX_WGRAD_FLOOR", "HX_BENCH_LD0", "HX_BENCH_NODB", "HX_BENCH_NOKFULL", "HX_SIM_CU_WORD",
                                "HX_COMM_TIMEOUT_S", "HX_COMM_INIT_TIMEOUT_S",
                                "HX_DIST_BACKEND", "HX_DP_FORCE_RCCL", "HX_BENCH_CHILD_PROBE", "HX_STEP_PROF", "HX_STEP_PROF_CHILD", "HX_REFERENCE_ROOT"};
  for (char** e = environ; e && *e; ++e) {
    if (strncmp(*e, "HX_", 3) != 0) continue;
    const char* eq = strchr(*e, '=');
    const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
    if (name.rfind("HX_EXTRA_FLAGS_", 0) == 0) continue;      // build-time flags of isaac_amd/build.py
    bool ok = false;
    for (const char* k : known) ok = ok || name == k;
    if (!ok) { hx_set_error("unknown environment variable " + name + ": this build reads no such knob (DESIGN.md 3.4 lists them)"); return -2; }
  }
  return 0;
}
int hx_knob_int(const char* name, int dflt, int lo, int hi, int* out) {
  *out = dflt;
  const char* e = getenv(name);
  if (!e) return 0;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if (end == e || *end != 0 || v < lo || v > hi) {
    hx_set_error(std::string(name) + "=" + e + ": expected an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]"); return -2;
  }
  *out = (int)v;
  return 0;
}
int hx_knob_hex32(const char* name, bool* present, unsigned* out) {
  const char* e = getenv(name);
  *present = e != nullptr;
  if (!e) return 0;
  char* end = nullptr;
  const unsigned long v = strtoul(e, &end, 16);
  if (end == e || *end != 0 || v > 0xfffffffful || v == 0) { hx_set_error(std::string(name) + "=" + e + ": expected a non-zero 32-bit hex CU mask word"); return -2; }
  *out = (unsigned)v;
  return 0;
}
extern "C" int hx_version(void) { return 100; }
#ifndef HX_BUILD_ID
#define HX_BUILD_ID "unknown"
#endif
extern "C" const char* hx_build_id(void) { return "100-" HX_BUILD_ID; }
extern "C" int hx_sync(void* stream) { HX_CHECK(hipStreamSynchronize((hipStream_t)stream)); return 0; }

struct hx_sim {
  int nd;                          // DoF count of the robot: 10 (hector) or 18 (hector with arms)
  int obs_f, priv_f, obs_ld, priv_ld, priv_stack;
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
  // frame-mode steps (hx_sim_step_frames): the bookkeeping of the last step, owed until its rows' next reader takes it
  hx_step_book book; bool book_owed;
  int* pause_word = nullptr;     // hx_sim_set_pause_word
  bool rows_stale = false;       // frame-mode steps have run since the row buffers were last written
  unsigned char* timeout_visible;
  long long step_counter;
  uint32_t rng_step;
  uint64_t seed;
  std::vector<void*> allocs;
  // HIP-event timing of the env-step kernel (hx_sim_time)
  bool timing = false; std::vector<hipEvent_t> ev; size_t ev_used = 0;
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
  if (int rc = hx_knobs_check()) return rc;      // before anything else: a mistyped knob must not run the default silently
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
  if (s->nd != HX_NUM_DOF && s->nd != HX_MAX_DOF && s->nd != HX_XBOT_DOF) { hx_set_error("hx_sim_create: num_dof must be 10 (hector), 18 (hector_full) or 12 (humanoid_ppo)"); return -2; }
  s->L = SLay(s->nd);
  s->obs_f = 11 + 3 * s->nd; s->priv_f = (s->nd == HX_XBOT_DOF ? 37 : 40) + 3 * s->nd;
  s->priv_stack = (s->nd == HX_XBOT_DOF) ? 3 : HX_FRAME_STACK;
  static_assert(HX_FRAME_STACK >= 3, "the privileged stack (15 or 3 frames) never exceeds the observation stack: the reset age saturates at HX_FRAME_STACK");
  if (s->priv_stack > HX_FRAME_STACK) { hx_set_error("hx_sim_create: priv_stack > HX_FRAME_STACK"); return -2; }
  s->obs_ld = (HX_FRAME_STACK * s->obs_f + 3) / 4 * 4; s->priv_ld = (s->priv_stack * s->priv_f + 3) / 4 * 4;
  s->seed = seed;
  s->step_counter = 0;
  s->rng_step = 0;
  s->cur = 0;
  if (int rc = hx_knobs_check()) return rc;
  bool cu_set = false; unsigned cu_word = 0;
  if (int rc = hx_knob_hex32("HX_SIM_CU_WORD", &cu_set, &cu_word)) return rc;
  if (stream) { s->stream = (hipStream_t)stream; s->own_stream = false; }
  else {
    if (cu_set) {      // experiment hook: confine this simulator's stream to a CU subset
      uint32_t mask[8];
      for (int i = 0; i < 8; ++i) mask[i] = cu_word;
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
  rc |= dalloc(s, &s->p.reset, n); rc |= dalloc(s, &s->p.age, n);
  s->book_owed = false;
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
    st[(size_t)SL_.BASE_MASS * n + e] = base_mass_h ? base_mass_h[e] : (s->nd == HX_NUM_DOF ? HXM_MASS0 : s->nd == HX_XBOT_DOF ? HXX_MASS0 : HXF_MASS0);
  }
  HX_CHECK(hipMemcpy(s->p.st, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice));
  // the env-step kernel keeps the height windows and the per-lane contact buffer in LDS: more than the 64 KB default
  HX_CHECK(hipFuncSetAttribute((const void*)hx_env_step_kernel<ModelHector>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HX_ENV_WPB * env_step_lds_bytes<ModelHector>())));
  if (HX_ENV_WPB * env_step_lds_bytes<ModelFull>() <= 160 * 1024) HX_CHECK(hipFuncSetAttribute((const void*)hx_env_step_kernel<ModelFull>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HX_ENV_WPB * env_step_lds_bytes<ModelFull>())));
  if (HX_ENV_WPB * env_step_lds_bytes<ModelXBot>() <= 160 * 1024) HX_CHECK(hipFuncSetAttribute((const void*)hx_env_step_kernel<ModelXBot>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(HX_ENV_WPB * env_step_lds_bytes<ModelXBot>())));
  if (dalloc(s, &s->cfg_d, 1)) return -3;
  HX_CHECK(hipMemcpy(s->cfg_d, &s->cfg, sizeof(hx_sim_cfg), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int hx_sim_set_terrain(hx_sim* s, const int16_t* heights_h, int32_t rows, int32_t cols, float horizontal_scale,
                                  float vertical_scale, float x0, float y0, float wall_height) {
  if (!s) { hx_set_error("hx_sim_set_terrain: null sim"); return -2; }
  if (!heights_h) { s->p.terrain = nullptr; return 0; }
  if (rows < HX_PATCH + 2 || cols < HX_PATCH + 2 || !(horizontal_scale > 0.f)) { hx_set_error("hx_sim_set_terrain: grid smaller than the contact window or bad scale"); return -2; }
  // metres in fp32, rounded from the double product exactly like the float32 mesh vertices of the reference
  std::vector<float> h((size_t)rows * cols);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((double)heights_h[i] * (double)vertical_scale);
  float* d = nullptr;
  HX_CHECK(hipMalloc((void**)&d, h.size() * sizeof(float)));
  s->allocs.push_back(d);
  HX_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  {
    const int prows = rows / 2, pcols = cols / 2;
    std::vector<float> pool((size_t)prows * pcols), poolw((size_t)prows * pcols);
    terrain_pool_build(h.data(), rows, cols, wall_height > 0.f ? wall_height : 0.f, pool.data(), poolw.data());
    float *dp = nullptr, *dw = nullptr;
    HX_CHECK(hipMalloc((void**)&dp, pool.size() * sizeof(float))); s->allocs.push_back(dp);
    HX_CHECK(hipMalloc((void**)&dw, poolw.size() * sizeof(float))); s->allocs.push_back(dw);
    HX_CHECK(hipMemcpy(dp, pool.data(), pool.size() * sizeof(float), hipMemcpyHostToDevice));
    HX_CHECK(hipMemcpy(dw, poolw.data(), poolw.size() * sizeof(float), hipMemcpyHostToDevice));
    s->p.t_pool = dp; s->p.t_poolw = dw; s->p.t_prows = prows; s->p.t_pcols = pcols;
  }
  s->p.terrain = d; s->p.t_rows = rows; s->p.t_cols = cols;
  s->p.t_hs = horizontal_scale; s->p.t_inv_hs = 1.0f / horizontal_scale; s->p.t_x0 = x0; s->p.t_y0 = y0;
  s->p.t_wall = wall_height > 0.f ? wall_height : 0.f;
  return 0;
}

extern "C" int hx_sim_set_terrain_options(hx_sim* s, int32_t flags) {
  if (!s) { hx_set_error("hx_sim_set_terrain_options: null sim"); return -2; }
  s->p.t_flags = flags;
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
  for (void* a : s->allocs) (void)hipFree(a);
  for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
  if (s->own_stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

struct StepOut { float* obs; float* priv; float* rew; unsigned char* done; unsigned char* timeout; };

// the bookkeeping of a frame-mode step that nobody took (hx_sim_take_book): one thread per robot
__global__ void __launch_bounds__(256) hx_book_kernel(hx_step_book b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < b.n) hx_step_book_row(b, e);
  if (blockIdx.x == 0 && threadIdx.x < 64) hx_step_book_global(b, (int)threadIdx.x);
}
static int flush_book(hx_sim* s) {
  if (!s->book_owed) return 0;
  hipLaunchKernelGGL(hx_book_kernel, dim3((s->cfg.num_envs + 255) / 256), dim3(256), 0, s->stream, s->book);
  s->book_owed = false;
  HX_CHECK(hipGetLastError());
  return 0;
}

// frames <-> rows (include/hx_sim.h): one workgroup per robot, coalesced along the row.
//   TO_FRAMES: frame p of the consumer's ring <- elements [p * f, (p + 1) * f) of the simulator's current row; kz from the age
//   else     : the simulator's row <- the window of `stack` frames starting at the slot, elements below kz read as zero
struct StackIoArgs { hx_frame_slot fs; float* obs_rows; float* priv_rows; const unsigned char* age; int n, obs_f, obs_ld, obs_stack, priv_f, priv_ld, priv_stack; };
template <bool TO_FRAMES> __global__ void __launch_bounds__(256) hx_stack_io_kernel(StackIoArgs a) {
  const int e = blockIdx.x;
  float* fo = a.fs.obs + (size_t)e * a.fs.obs_env_stride;
  float* fp = a.fs.priv + (size_t)e * a.fs.priv_env_stride;
  float* ro = a.obs_rows + (size_t)e * a.obs_ld;
  float* rp = a.priv_rows + (size_t)e * a.priv_ld;
  const int wo = a.obs_stack * a.obs_f, wp = a.priv_stack * a.priv_f;
  if (TO_FRAMES) {
    for (int k = threadIdx.x; k < wo; k += blockDim.x) fo[k] = ro[k];
    for (int k = threadIdx.x; k < wp; k += blockDim.x) fp[k] = rp[k];
    if (threadIdx.x == 0) {
      const int age = a.age[e];
      a.fs.obs_kz[e] = (a.obs_stack - min(age, a.obs_stack)) * a.obs_f;
      a.fs.priv_kz[e] = (a.priv_stack - min(age, a.priv_stack)) * a.priv_f;
    }
  } else {
    const int zo = a.fs.obs_kz[e], zp = a.fs.priv_kz[e];
    for (int k = threadIdx.x; k < a.obs_ld; k += blockDim.x) ro[k] = (k >= zo && k < wo) ? fo[k] : 0.f;
    for (int k = threadIdx.x; k < a.priv_ld; k += blockDim.x) rp[k] = (k >= zp && k < wp) ? fp[k] : 0.f;
  }
}

static int launch_step(hx_sim* s, const float* actions, const float* pack, int mode, const StepOut* out, const hx_frame_slot* frames = nullptr) {
  const int n = s->cfg.num_envs;
  { const int rc = flush_book(s); if (rc) return rc; }      // a frame-mode step whose bookkeeping nobody took
  StepArgs A{};
  const int env_blocks = ((n + HX_RPW - 1) / HX_RPW + HX_ENV_WPB - 1) / HX_ENV_WPB;       // workgroups of HX_ENV_WPB waves
  A.frames = frames != nullptr; A.obs_stack = HX_FRAME_STACK; A.priv_stack = s->priv_stack; A.clip = s->cfg.clip_observations;
  if (frames) A.fs = *frames;
  A.mode = mode;
  if (mode == 0) s->step_counter += 1;
  A.step_counter = s->step_counter;
  A.k0 = (uint32_t)(s->seed & 0xffffffffu);
  A.k1 = (uint32_t)(s->seed >> 32);
  A.rng_step = s->rng_step++;
  // reset counter: ping-pong pair; the bookkeeping of step t zeroes the counter step t+1 will use
  s->p.num_reset = s->num_reset2[s->parity];
  bool timed = s->timing && mode == 0;
  if (timed) {
    while (s->ev_used + 2 > s->ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; s->ev.push_back(e); }
    timed = s->ev_used + 2 <= s->ev.size();
  }
  if (timed) (void)hipEventRecord(s->ev[s->ev_used], s->stream);
  if (s->nd == HX_XBOT_DOF) hipLaunchKernelGGL(hx_env_step_kernel<ModelXBot>, dim3(env_blocks), dim3(64 * HX_ENV_WPB), HX_ENV_WPB * env_step_lds_bytes<ModelXBot>(), s->stream, s->p, s->cfg_d, actions, pack, A);
  else if (s->nd == HX_NUM_DOF) hipLaunchKernelGGL(hx_env_step_kernel<ModelHector>, dim3(env_blocks), dim3(64 * HX_ENV_WPB), HX_ENV_WPB * env_step_lds_bytes<ModelHector>(), s->stream, s->p, s->cfg_d, actions, pack, A);
  else hipLaunchKernelGGL(hx_env_step_kernel<ModelFull>, dim3(env_blocks), dim3(64 * HX_ENV_WPB), HX_ENV_WPB * env_step_lds_bytes<ModelFull>(), s->stream, s->p, s->cfg_d, actions, pack, A);
  if (timed) { (void)hipEventRecord(s->ev[s->ev_used + 1], s->stream); s->ev_used += 2; }
  if (frames) {
    // no stacking launch and no rows: the step's bookkeeping travels to the next reader of the frames
    s->book = hx_step_book{s->p.reset, s->p.timeout, s->timeout_visible, s->num_reset2[s->parity], s->num_reset2[s->parity ^ 1], s->p.stat_sum, s->p.stat_last,
                           s->p.stat_acc, s->p.stat_steps, s->p.rew, out ? out->rew : nullptr, out ? out->done : nullptr, out ? out->timeout : nullptr, n};
    s->book_owed = true;
    s->rows_stale = true;
    s->parity ^= 1;
    HX_CHECK(hipGetLastError());
    return 0;
  }
  if (s->rows_stale) { hx_set_error("hx_sim_step: the simulator's row buffers are stale after frame-mode steps (call hx_sim_import_stack first)"); return -2; }
  // destination of the new observation rows: the caller's (learner storage) or the other internal buffer
  float* od = s->obs[s->cur ^ 1]; float* pd = s->priv[s->cur ^ 1];
  if (s->obs_cur == od) { od = s->obs[s->cur]; pd = s->priv[s->cur]; }
  if (out && out->obs) {
    od = out->obs; pd = out->priv;
    // the new rows are built from the previous step's (one frame down): in place is a race between the launch's workgroups
    if (od == s->obs_cur || pd == s->priv_cur) { hx_set_error("hx_sim_step_ex: obs_dst / priv_dst are the rows of the previous step (HX_BUF_OBS / HX_BUF_PRIV): the new rows need a buffer of their own"); return -2; }
  } else s->cur ^= 1;
  StackArgs k{};
  k.obs_src = s->obs_cur; k.obs_dst = od; k.obs_frame = s->p.obs_frame;
  k.priv_src = s->priv_cur; k.priv_dst = pd; k.priv_frame = s->p.priv_frame;
  k.reset = s->p.reset; k.timeout = s->p.timeout; k.timeout_visible = s->timeout_visible;
  k.num_reset = s->num_reset2[s->parity]; k.num_reset_next = s->num_reset2[s->parity ^ 1];
  k.stat_sum = s->p.stat_sum; k.stat_last = s->p.stat_last; k.stat_acc = s->p.stat_acc; k.stat_steps = s->p.stat_steps;
  k.rew = s->p.rew; k.rew_out = out ? out->rew : nullptr; k.done_out = out ? out->done : nullptr; k.timeout_out = out ? out->timeout : nullptr;
  k.n = n; k.clip = s->cfg.clip_observations; k.pause = (out && out->obs) ? s->pause_word : nullptr;
  k.obs_f = s->obs_f; k.obs_ld = s->obs_ld; k.priv_f = s->priv_f; k.priv_ld = s->priv_ld; k.priv_stack = s->priv_stack;
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
  if ((((uintptr_t)obs_dst) | ((uintptr_t)priv_dst)) & 15) { hx_set_error("hx_sim_step_ex: obs_dst / priv_dst must be 16-byte aligned (the rows are written with 16-byte stores)"); return -2; }
  StepOut o{obs_dst, priv_dst, rew_dst, done_dst, timeout_dst};
  return launch_step(s, actions, pack, 0, &o);
}

extern "C" int hx_sim_set_pause_word(hx_sim* s, int32_t* word) {
  if (!s) { hx_set_error("hx_sim_set_pause_word: null sim"); return -2; }
  s->pause_word = word;
  return 0;
}

// ---- single-frame observation storage (include/hx_sim.h)
extern "C" int hx_sim_step_frames(hx_sim* s, const float* actions, const float* pack, const hx_frame_slot* dst,
                                  float* rew_dst, uint8_t* done_dst, uint8_t* timeout_dst) {
  if (!s || !actions || !dst || !dst->obs || !dst->priv || !dst->obs_kz || !dst->priv_kz) { hx_set_error("hx_sim_step_frames: null argument"); return -2; }
  StepOut o{nullptr, nullptr, rew_dst, done_dst, timeout_dst};
  return launch_step(s, actions, pack, 0, &o, dst);
}
extern "C" int hx_sim_take_book(hx_sim* s, hx_step_book* book, int32_t* valid) {
  if (!s || !book || !valid) { hx_set_error("hx_sim_take_book: null argument"); return -2; }
  *valid = s->book_owed ? 1 : 0;
  if (s->book_owed) *book = s->book;
  s->book_owed = false;
  return 0;
}
extern "C" int hx_sim_flush_book(hx_sim* s) {
  if (!s) { hx_set_error("hx_sim_flush_book: null sim"); return -2; }
  return flush_book(s);
}
static StackIoArgs stack_io_args(hx_sim* s, const hx_frame_slot* f) {
  StackIoArgs a{};
  a.fs = *f; a.obs_rows = s->obs_cur; a.priv_rows = s->priv_cur; a.age = s->p.age; a.n = s->cfg.num_envs;
  a.obs_f = s->obs_f; a.obs_ld = s->obs_ld; a.obs_stack = HX_FRAME_STACK; a.priv_f = s->priv_f; a.priv_ld = s->priv_ld; a.priv_stack = s->priv_stack;
  return a;
}
extern "C" int hx_sim_export_stack(hx_sim* s, const hx_frame_slot* first) {
  if (!s || !first || !first->obs || !first->priv || !first->obs_kz || !first->priv_kz) { hx_set_error("hx_sim_export_stack: null argument"); return -2; }
  if (s->rows_stale) { hx_set_error("hx_sim_export_stack: the simulator's rows are stale (frame-mode steps since the last hx_sim_import_stack)"); return -2; }
  hipLaunchKernelGGL(hx_stack_io_kernel<true>, dim3(s->cfg.num_envs), dim3(256), 0, s->stream, stack_io_args(s, first));
  HX_CHECK(hipGetLastError());
  return 0;
}
extern "C" int hx_sim_import_stack(hx_sim* s, const hx_frame_slot* first) {
  if (!s || !first || !first->obs || !first->priv || !first->obs_kz || !first->priv_kz) { hx_set_error("hx_sim_import_stack: null argument"); return -2; }
  // rows land in the simulator's own buffers (not in a caller's storage a former hx_sim_step_ex pointed obs_cur at)
  s->obs_cur = s->obs[s->cur]; s->priv_cur = s->priv[s->cur];
  hipLaunchKernelGGL(hx_stack_io_kernel<false>, dim3(s->cfg.num_envs), dim3(256), 0, s->stream, stack_io_args(s, first));
  s->rows_stale = false;
  HX_CHECK(hipGetLastError());
  return 0;
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
extern "C" int64_t hx_sim_step_counter(hx_sim* s) { return s ? (int64_t)s->step_counter : -1; }

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
// HIP-event timing of the env-step kernel on the simulator's stream: which = 1 start / clear, 0 stop and read
// {total milliseconds, launches} (bench.py: the live duration behind roofline.env_step)
extern "C" int hx_sim_time(hx_sim* s, int which, double* out_h /*[2]*/) {
  if (!s) { hx_set_error("hx_sim_time: null sim"); return -2; }
  if (which == 1) { s->timing = true; s->ev_used = 0; return 0; }
  s->timing = false;
  HX_CHECK(hipStreamSynchronize(s->stream));
  double ms = 0;
  for (size_t i = 0; i + 1 < s->ev_used; i += 2) { float t = 0; HX_CHECK(hipEventElapsedTime(&t, s->ev[i], s->ev[i + 1])); ms += t; }
  if (out_h) { out_h[0] = ms; out_h[1] = (double)(s->ev_used / 2); }
  s->ev_used = 0;
  return 0;
}
extern "C" void* hx_sim_stream(hx_sim* s) { return (void*)s->stream; }
// measurement hook (tools/step_prof.py; library built with -DHX_STEP_PROF): which = 1 starts / clears, 0 reads the cycle
// counters summed over all waves and launches since: {window fetch + pooling, action processing, kinematics, contact
// phase, articulated inertias, exchange + base solve, accelerations + forces + integration, guard + gather, glue}
extern "C" int hx_sim_prof(hx_sim* s, int which, long long* out_h /*[18]*/) {
  if (!s) { hx_set_error("hx_sim_prof: null sim"); return -2; }
  if (which == 1) {
    if (!s->p.prof) { long long* d = nullptr; if (dalloc(s, &d, 20 + 19 * HX_PROF_WAVES)) return -3; s->p.prof = d; }
    HX_CHECK(hipStreamSynchronize(s->stream));
    HX_CHECK(hipMemset(s->p.prof, 0, (20 + 19 * HX_PROF_WAVES) * sizeof(long long)));
    return 0;
  }
  if (!s->p.prof || !out_h) { hx_set_error("hx_sim_prof: not started"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(out_h, s->p.prof, 18 * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
// lifetimes of the first `n` env-step waves (100 MHz ticks, summed over the launches since hx_sim_prof(s, 1, ..)); -DHX_STEP_PROF builds
extern "C" int hx_sim_prof_waves(hx_sim* s, long long* out_h, int n) {
  if (!s || !s->p.prof || !out_h || n < 1 || n > HX_PROF_WAVES) { hx_set_error("hx_sim_prof_waves: not started or bad count"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  HX_CHECK(hipMemcpy(out_h, s->p.prof + 20, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
// of the most recent env-step launch, for its first `n` waves: out_h[0][n] start and [1][n] end (100 MHz ticks of the device-wide
// counter), [2][n] (XCC_ID << 32) | HW_ID of the SIMD the wave ran on, [3..11][n] the wave's cycles in the nine phases of
// hx_sim_prof, [12][n] shapes visited by its contact loop (all substeps), [13..17][n] visits of the first five shapes;
// -DHX_STEP_PROF builds (tools/env_waves.py)
extern "C" int hx_sim_prof_last(hx_sim* s, long long* out_h, int n) {
  if (!s || !s->p.prof || !out_h || n < 1 || n > HX_PROF_WAVES) { hx_set_error("hx_sim_prof_last: not started or bad count"); return -2; }
  HX_CHECK(hipStreamSynchronize(s->stream));
  for (int k = 0; k < 18; ++k)
    HX_CHECK(hipMemcpy(out_h + (size_t)k * n, s->p.prof + 20 + (size_t)(k + 1) * HX_PROF_WAVES, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}
