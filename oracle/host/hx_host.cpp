// hx_host.cpp -- HOST (CPU, OpenMP) build of the simulator's single-source device code.
//
// TEST INFRASTRUCTURE, NEVER SHIPPED (see oracle/__init__.py): it exists so that
//   * the exact text of the kernels' math (isaac_amd/csrc/hx_math.h, hx_dyn.h, hx_env.h) can be compared with the numpy
//     oracle in the CPU test suite, without a GPU, and run under -fsanitize=address,undefined;
//   * bench.py's cpu_baseline leg can time the same env step on the GPU box's host cores, OpenMP over robots, next to the
//     reference's PhysX thread count (physx.num_threads = 10, hector_config.py:109).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load the library built from this file; the
// product (isaac_amd/) never does and fails loudly without the HIP library.
// Build: oracle/host/Makefile (g++ -O3 -march=native -fopenmp; `make asan` for the sanitizer build).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../isaac_amd/csrc/hx_env.h"

struct hxh_env {
  int nd;
  hx_sim_cfg cfg;
  SimPtrs p;
  SLay L{10};
  int obs_f, priv_f, obs_ld, priv_ld, priv_stack;
  std::vector<float> tbl;                 // what the kernel stages in LDS: side tables, base table, PD constants
  std::vector<float> obs[2], priv[2];
  int cur;
  std::vector<unsigned char> timeout_visible;
  long long step_counter;
  uint32_t rng_step;
  uint64_t seed;
  std::vector<std::vector<char>> bufs;
  std::vector<float> terrain, tpool, tpoolw;
};

template <typename T> static T* halloc(hxh_env* s, size_t count) {
  s->bufs.emplace_back(count * sizeof(T), 0);
  return reinterpret_cast<T*>(s->bufs.back().data());
}

template <class M> static void stage(hxh_env* s) {
  s->tbl.assign(ModelInfo<M>::LDS_FLOATS, 0.f);
  dyn_stage_constants<M>(s->tbl.data(), 0, 1, s->cfg.p_gains, s->cfg.d_gains, s->cfg.torque_limits, s->cfg.default_dof_pos);
}

extern "C" hxh_env* hxh_create(const hx_sim_cfg* cfg, const float* friction, const float* base_mass, const float* origins, const float* start_pos,
                               uint64_t seed) {
  hxh_env* s = new hxh_env();
  s->cfg = *cfg;
  s->nd = cfg->num_dof ? cfg->num_dof : HX_NUM_DOF;
  if (s->nd != 10 && s->nd != 18 && s->nd != 12) { delete s; return nullptr; }
  s->L = SLay(s->nd);
  s->obs_f = 11 + 3 * s->nd; s->priv_f = (s->nd == 12 ? 37 : 40) + 3 * s->nd;
  s->priv_stack = s->nd == 12 ? 3 : HX_FRAME_STACK;
  s->obs_ld = (HX_FRAME_STACK * s->obs_f + 3) / 4 * 4; s->priv_ld = (s->priv_stack * s->priv_f + 3) / 4 * 4;
  s->seed = seed; s->step_counter = 0; s->rng_step = 0; s->cur = 0;
  const size_t n = cfg->num_envs;
  SimPtrs& p = s->p;
  p = SimPtrs{};
  p.st = halloc<float>(s, (size_t)s->L.SIZE * n);
  p.ep_len = halloc<int>(s, n);
  p.ep_sums = halloc<float>(s, (size_t)HX_NUM_REWARDS * n);
  p.torques = halloc<float>(s, (size_t)s->nd * n);
  p.contact = halloc<float>(s, (size_t)(1 + s->nd) * 3 * n);
  p.bodies = halloc<float>(s, 52 * n);
  p.obs_frame = halloc<float>(s, (size_t)s->obs_f * n);
  p.priv_frame = halloc<float>(s, (size_t)s->priv_f * n);
  p.rew = halloc<float>(s, n);
  p.reset = halloc<unsigned char>(s, n);
  p.age = halloc<unsigned char>(s, n);
  p.timeout = halloc<unsigned char>(s, n);
  p.num_reset = halloc<int>(s, 1);
  p.stat_sum = halloc<float>(s, HX_NUM_REWARDS); p.stat_last = halloc<float>(s, HX_NUM_REWARDS); p.stat_acc = halloc<float>(s, HX_NUM_REWARDS);
  p.stat_steps = halloc<int>(s, 2); p.stat_ring = halloc<float>(s, 2 * HX_STAT_RING); p.stat_cnt = halloc<int>(s, 2);
  s->timeout_visible.assign(n, 0);
  for (int i = 0; i < 2; ++i) { s->obs[i].assign(n * s->obs_ld, 0.f); s->priv[i].assign(n * s->priv_ld, 0.f); }
  const SLay& SL_ = s->L;
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 3; ++k) {
      p.st[(size_t)(SL_.ROOT_POS + k) * n + e] = start_pos ? start_pos[e * 3 + k] : 0.f;
      p.st[(size_t)(SL_.ORIGIN + k) * n + e] = origins ? origins[e * 3 + k] : 0.f;
    }
    p.st[(size_t)(SL_.ROOT_QUAT + 3) * n + e] = 1.f;
    p.st[(size_t)SL_.LAST_FEET_Z * n + e] = 0.05f;
    p.st[(size_t)(SL_.LAST_FEET_Z + 1) * n + e] = 0.05f;
    p.st[(size_t)SL_.FRICTION * n + e] = friction ? friction[e] : 1.f;
    p.st[(size_t)SL_.BASE_MASS * n + e] = base_mass ? base_mass[e] : (s->nd == 10 ? ModelHector::MASS0 : s->nd == 12 ? ModelXBot::MASS0 : ModelFull::MASS0);
  }
  if (s->nd == 10) stage<ModelHector>(s); else if (s->nd == 12) stage<ModelXBot>(s); else stage<ModelFull>(s);
  return s;
}

extern "C" void hxh_destroy(hxh_env* s) { delete s; }

extern "C" int hxh_set_terrain(hxh_env* s, const int16_t* heights, int rows, int cols, float hs, float vs, float x0, float y0, float wall) {
  if (!heights) { s->p.terrain = nullptr; return 0; }
  if (rows < HX_PATCH + 2 || cols < HX_PATCH + 2) return -2;
  s->terrain.resize((size_t)rows * cols);
  for (size_t i = 0; i < s->terrain.size(); ++i) s->terrain[i] = (float)((double)heights[i] * (double)vs);
  s->tpool.assign((size_t)(rows / 2) * (cols / 2), 0.f); s->tpoolw.assign((size_t)(rows / 2) * (cols / 2), 0.f);
  terrain_pool_build(s->terrain.data(), rows, cols, wall > 0.f ? wall : 0.f, s->tpool.data(), s->tpoolw.data());
  s->p.t_pool = s->tpool.data(); s->p.t_poolw = s->tpoolw.data(); s->p.t_prows = rows / 2; s->p.t_pcols = cols / 2;
  s->p.terrain = s->terrain.data(); s->p.t_rows = rows; s->p.t_cols = cols;
  s->p.t_hs = hs; s->p.t_inv_hs = 1.0f / hs; s->p.t_x0 = x0; s->p.t_y0 = y0; s->p.t_wall = wall;
  return 0;
}

template <class M> static void step_robot(hxh_env* s, const float* actions, const float* pack, const StepArgs& A, int e) {
  using D = TaskDims<M>;
  using MI = ModelInfo<M>;
  constexpr int NL = M::NL, ND = 2 * NL;
  const hx_sim_cfg& cfg = s->cfg;
  const SimPtrs& p = s->p;
  const int n = cfg.num_envs;
  const SLay SL(ND);
  auto LD = [&](int f) { return p.st[(size_t)f * n + e]; };
  SideConst<M> C[2]; C[0].bind(s->tbl.data(), 0); C[1].bind(s->tbl.data(), 1);
  Rng rng; rng.pack = pack; rng.share = nullptr; rng.lane = 0; rng.nlanes = 1; rng.n = n; rng.env = e; rng.gid = (uint32_t)(e + cfg.env_id_offset);
  rng.k0 = A.k0; rng.k1 = A.k1; rng.step = A.rng_step;
  DynStateT<M> S[2];
  for (int sd = 0; sd < 2; ++sd) {
    S[sd].pos = mk(LD(SL.ROOT_POS), LD(SL.ROOT_POS + 1), LD(SL.ROOT_POS + 2));
    for (int i = 0; i < 4; ++i) S[sd].quat[i] = LD(SL.ROOT_QUAT + i);
    S[sd].linvel = mk(LD(SL.LINVEL), LD(SL.LINVEL + 1), LD(SL.LINVEL + 2));
    S[sd].angvel = mk(LD(SL.ANGVEL), LD(SL.ANGVEL + 1), LD(SL.ANGVEL + 2));
    for (int j = 0; j < NL; ++j) { S[sd].q[j] = LD(SL.Q + sd * NL + j); S[sd].qd[j] = LD(SL.QD + sd * NL + j); }
  }
  RobotVals<M> R;
  for (int j = 0; j < ND; ++j) R.act[j] = LD(SL.ACT + j);
  R.friction = LD(SL.FRICTION); R.base_mass = LD(SL.BASE_MASS);
  R.ep_len = p.ep_len[e];
  R.blown = false;
  float tau[2][NL];
  for (int sd = 0; sd < 2; ++sd) for (int j = 0; j < NL; ++j) tau[sd][j] = 0.f;
  SideForcesT<M> F[2];
  for (int sd = 0; sd < 2; ++sd) { F[sd].base = mk(0, 0, 0); for (int q = 0; q < MI::NSHAPE; ++q) F[sd].shape[q] = mk(0, 0, 0); }
  if (A.mode == 0) {
    env_actions<M>(cfg, rng, actions + (size_t)e * ND, R.act);
    DynParams P = dyn_params(cfg, R.friction);
    float patch[HX_PATCH * HX_PATCH], pool[HX_POOL * HX_POOL], poolw[HX_POOL * HX_POOL];
    if (p.terrain != nullptr) {
      int oi, oj;
      patch_origin(p, S[0].pos.x, S[0].pos.y, oi, oj);
      for (int idx = 0; idx < HX_PATCH * HX_PATCH; ++idx) patch[idx] = p.terrain[(size_t)(oi + idx / HX_PATCH) * p.t_cols + (oj + idx % HX_PATCH)];
      P.patch = patch; P.px0 = p.t_x0 + (float)oi * p.t_hs; P.py0 = p.t_y0 + (float)oj * p.t_hs; P.inv_hs = p.t_inv_hs; P.wall = p.t_wall;
      for (int c = 0; c < HX_POOL * HX_POOL; ++c) {
        const size_t g = (size_t)(oi / 2 + c / HX_POOL) * p.t_pcols + (oj / 2 + c % HX_POOL);
        pool[c] = p.t_pool[g]; poolw[c] = p.t_poolw[g];
      }
      P.pool = pool; P.poolw = poolw;
    }
    float target[2][NL];
    for (int sd = 0; sd < 2; ++sd)
      for (int j = 0; j < NL; ++j) target[sd][j] = R.act[sd * NL + j] * cfg.action_scale + cfg.default_dof_pos[sd * NL + j];
    const float mass_scale = R.base_mass / M::MASS0;
    for (int sub = 0; sub < cfg.decimation; ++sub)
      dyn_substep_pair<M>(S[0], S[1], P, C[0], C[1], target[0], target[1], mass_scale, tau[0], tau[1], sub == cfg.decimation - 1, F[0], F[1]);
    if (dyn_state_bad<M>(S[0]) || dyn_state_bad<M>(S[1])) {
      R.blown = true;
      for (int sd = 0; sd < 2; ++sd) {
        S[sd].pos = mk(cfg.base_init_state[0] + LD(SL.ORIGIN), cfg.base_init_state[1] + LD(SL.ORIGIN + 1), cfg.base_init_state[2] + LD(SL.ORIGIN + 2));
        for (int k = 0; k < 4; ++k) S[sd].quat[k] = cfg.base_init_state[3 + k];
        S[sd].linvel = mk(0, 0, 0); S[sd].angvel = mk(0, 0, 0);
        for (int j = 0; j < NL; ++j) { S[sd].q[j] = cfg.default_dof_pos[sd * NL + j]; S[sd].qd[j] = 0.f; tau[sd][j] = 0.f; }
        F[sd].base = mk(0, 0, 0);
        for (int q = 0; q < MI::NSHAPE; ++q) F[sd].shape[q] = mk(0, 0, 0);
      }
    }
  }
  for (int sd = 0; sd < 2; ++sd) dyn_body_states<M>(S[sd], C[sd], R.bo[2 * sd], R.bo[2 * sd + 1]);
  R.pos = S[0].pos; for (int k = 0; k < 4; ++k) R.quat[k] = S[0].quat[k];
  R.linvel = S[0].linvel; R.angvel = S[0].angvel;
  for (int sd = 0; sd < 2; ++sd)
    for (int j = 0; j < NL; ++j) { R.qa[sd * NL + j] = S[sd].q[j]; R.qda[sd * NL + j] = S[sd].qd[j]; R.torques[sd * NL + j] = tau[sd][j]; }
  R.f_base = F[0].base;
  for (int sd = 0; sd < 2; ++sd) for (int q = 0; q < MI::NSHAPE; ++q) R.side_force[sd][q] = F[sd].shape[q];
  env_glue<M>(p, cfg, A, n, e, true, rng, R);
}

static void stack_frames(hxh_env* s) {
  const int n = s->cfg.num_envs;
  const float clip = s->cfg.clip_observations;
  const std::vector<float>&so = s->obs[s->cur], &sp = s->priv[s->cur];
  std::vector<float>&dob = s->obs[s->cur ^ 1], &dpr = s->priv[s->cur ^ 1];
  const int nr = *s->p.num_reset;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    const bool rst = s->p.reset[e] != 0;
    for (int stream = 0; stream < 2; ++stream) {
      const int F = stream ? s->priv_f : s->obs_f, ld = stream ? s->priv_ld : s->obs_ld, keep = ((stream ? s->priv_stack : HX_FRAME_STACK) - 1) * F;
      const float* src = (stream ? sp.data() : so.data()) + (size_t)e * ld;
      float* dst = (stream ? dpr.data() : dob.data()) + (size_t)e * ld;
      const float* fr = stream ? s->p.priv_frame : s->p.obs_frame;
      for (int k = 0; k < ld; ++k) {
        float v = 0.f;
        if (k < keep) v = rst ? 0.f : src[k + F];
        else if (k < keep + F) v = fminf(fmaxf(fr[(size_t)e * F + (k - keep)], -clip), clip);
        dst[k] = v;
      }
    }
    if (nr > 0) s->timeout_visible[e] = s->p.timeout[e];
  }
  if (nr > 0) {
    for (int r = 0; r < HX_NUM_REWARDS; ++r) { s->p.stat_last[r] = s->p.stat_sum[r] / (float)nr; s->p.stat_sum[r] = 0.f; }
    s->p.stat_steps[1] = 1;
  }
  if (s->p.stat_steps[1]) { for (int r = 0; r < HX_NUM_REWARDS; ++r) s->p.stat_acc[r] += s->p.stat_last[r]; s->p.stat_steps[0] += 1; }
  s->cur ^= 1;
}

static void run_step(hxh_env* s, const float* actions, const float* pack, int mode) {
  StepArgs A{};
  A.frames = 0; A.obs_stack = HX_FRAME_STACK; A.priv_stack = HX_FRAME_STACK; A.clip = s->cfg.clip_observations;
  A.mode = mode;
  if (mode == 0) s->step_counter += 1;
  A.step_counter = s->step_counter;
  A.k0 = (uint32_t)(s->seed & 0xffffffffu); A.k1 = (uint32_t)(s->seed >> 32);
  A.rng_step = s->rng_step++;
  *s->p.num_reset = 0;
  const int n = s->cfg.num_envs;
#pragma omp parallel for schedule(dynamic, 16)
  for (int e = 0; e < n; ++e) {
    if (s->nd == 10) step_robot<ModelHector>(s, actions, pack, A, e);
    else if (s->nd == 12) step_robot<ModelXBot>(s, actions, pack, A, e);
    else step_robot<ModelFull>(s, actions, pack, A, e);
  }
  stack_frames(s);
}

extern "C" void hxh_reset_all(hxh_env* s, const float* pack) { run_step(s, nullptr, pack, 1); }
extern "C" void hxh_step(hxh_env* s, const float* actions, const float* pack) { run_step(s, actions, pack, 0); }

extern "C" void* hxh_buffer(hxh_env* s, int which) {
  const size_t n = s->cfg.num_envs;
  switch (which) {
    case HX_BUF_OBS: return s->obs[s->cur].data();
    case HX_BUF_PRIV: return s->priv[s->cur].data();
    case HX_BUF_REW: return s->p.rew;
    case HX_BUF_RESET: return s->p.reset;
    case HX_BUF_TIMEOUT: return s->p.timeout;
    case HX_BUF_TIMEOUT_VISIBLE: return s->timeout_visible.data();
    case HX_BUF_EP_LEN: return s->p.ep_len;
    case HX_BUF_COMMANDS: return s->p.st + (size_t)s->L.CMD * n;
    case HX_BUF_TORQUES: return s->p.torques;
    case HX_BUF_CONTACT: return s->p.contact;
    case HX_BUF_BODY_STATE: return s->p.bodies;
    case HX_BUF_EPISODE_SUMS: return s->p.ep_sums;
    case HX_BUF_FEET_AIR_TIME: return s->p.st + (size_t)s->L.AIR * n;
    case HX_BUF_FEET_HEIGHT: return s->p.st + (size_t)s->L.FEET_H * n;
    case HX_BUF_NUM_RESET: return s->p.num_reset;
    default: return nullptr;
  }
}
extern "C" void* hxh_state(hxh_env* s) { return s->p.st; }            // [SLay.SIZE][N], field-major
extern "C" int hxh_state_size(hxh_env* s) { return s->L.SIZE; }
extern "C" void hxh_get_state(hxh_env* s, float* root, float* q, float* qd) {
  const size_t n = s->cfg.num_envs; const int nd = s->nd;
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 13; ++k) root[e * 13 + k] = s->p.st[(size_t)k * n + e];
    for (int j = 0; j < nd; ++j) { q[e * nd + j] = s->p.st[(size_t)(s->L.Q + j) * n + e]; qd[e * nd + j] = s->p.st[(size_t)(s->L.QD + j) * n + e]; }
  }
}
extern "C" void hxh_set_state(hxh_env* s, const float* root, const float* q, const float* qd) {
  const size_t n = s->cfg.num_envs; const int nd = s->nd;
  for (size_t e = 0; e < n; ++e) {
    for (int k = 0; k < 13; ++k) s->p.st[(size_t)k * n + e] = root[e * 13 + k];
    for (int j = 0; j < nd; ++j) { s->p.st[(size_t)(s->L.Q + j) * n + e] = q[e * nd + j]; s->p.st[(size_t)(s->L.QD + j) * n + e] = qd[e * nd + j]; }
  }
}
extern "C" void hxh_set_episode_length(hxh_env* s, const int32_t* h) { for (int e = 0; e < s->cfg.num_envs; ++e) s->p.ep_len[e] = h[e]; }
extern "C" void hxh_set_step_counter(hxh_env* s, int64_t c) { s->step_counter = c; }
extern "C" void hxh_set_commands(hxh_env* s, const float* cmd) {
  const size_t n = s->cfg.num_envs;
  for (size_t e = 0; e < n; ++e) for (int k = 0; k < 4; ++k) s->p.st[(size_t)(s->L.CMD + k) * n + e] = cmd[e * 4 + k];
}
extern "C" int hxh_num_threads(void) {
  int t = 1;
#pragma omp parallel
  {
#pragma omp single
    t =
#ifdef _OPENMP
        __builtin_omp_get_num_threads();
#else
        1;
#endif
  }
  return t;
}
