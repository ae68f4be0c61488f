/* hx_sim.h -- C ABI of the MI355X-native hector environment step.
 *
 * The reference has no FFI for this path: its seams are two Python protocols.  This
 * header is the C boundary a maintainer binds instead of them.  Each entry point names the reference
 * interface it replaces:
 *
 *   hx_sim_create        LeggedRobot.__init__ / _create_envs / _init_buffers
 *                        (humanoid/envs/base/legged_robot.py:58-82,433-515,587-681) and the
 *                        gym.create_sim / load_asset / create_actor / prepare_sim calls underneath
 *   hx_sim_reset_all     HectorFreeEnv.__init__ tail: reset_idx(all) + compute_observations()
 *                        (humanoid/envs/custom/hector_env.py:50-51)
 *   hx_sim_step          HectorFreeEnv.step -> LeggedRobot.step -> post_physics_step
 *                        (hector_env.py:158-169, legged_robot.py:84-153), including the 10 x
 *                        {_compute_torques, gym.set_dof_actuation_force_tensor, gym.simulate,
 *                        gym.refresh_dof_state_tensor} substeps (legged_robot.py:93-100)
 *   hx_sim_buffer        VecEnv attributes obs_buf / privileged_obs_buf / rew_buf / reset_buf /
 *                        episode_length_buf / extras["time_outs"] (humanoid/algo/vec_env.py:37-61)
 *   hx_sim_get_state /   gym.acquire_actor_root_state_tensor / acquire_dof_state_tensor and
 *   hx_sim_set_state     gym.set_actor_root_state_tensor[_indexed] / set_dof_state_tensor_indexed
 *                        (legged_robot.py:437-456,370,394)
 *   hx_sim_set_episode_length   `env.episode_length_buf = randint_like(...)`
 *                        (humanoid/algo/ppo/on_policy_runner.py:103-106)
 *   hx_sim_episode_stats extras["episode"]["rew_*"] (legged_robot.py:198-201)
 *
 * Conventions: single-threaded caller; every call enqueues on the one hipStream_t given at creation
 * and returns without synchronising unless it copies to host; pointers are DEVICE pointers unless
 * the name ends in _h; the library owns all device buffers.  Return value 0 = ok, negative = error,
 * message from hx_last_error().  No CPU fallback exists: without a HIP device every call fails.
 */
#ifndef HX_SIM_H
#define HX_SIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HX_NUM_DOF 10          /* hector */
#define HX_MAX_DOF 18          /* hector_full (legs + arms, hector_w_arm_config.py:17); selected by hx_sim_cfg.num_dof */
#define HX_XBOT_DOF 12         /* humanoid_ppo (XBot-L, humanoid_config.py:46): frames 47 / 73, privileged rows 3 frames deep */
#define HX_MAX_OBS_FRAME 65    /* 11 + 3 * 18 (hector_w_arm_config.py:12) */
#define HX_NUM_BODIES 11
#define HX_OBS_FRAME 41        /* num_single_obs, hector_config.py:12 */
#define HX_PRIV_FRAME 70       /* single_num_privileged_obs, hector_config.py:14 */
#define HX_FRAME_STACK 15      /* frame_stack = c_frame_stack, hector_config.py:10-11 */
#define HX_NUM_OBS 615
#define HX_NUM_PRIV 1050
#define HX_OBS_LD 616          /* row stride of the obs buffer (16-byte aligned rows) */
#define HX_PRIV_LD 1052        /* row stride of the privileged obs buffer */
#define HX_NUM_REWARDS 22      /* every _reward_* of hector_env.py, alphabetical (helpers.py:47 dir() order) */

/* Random pack: one step's injected random numbers, laid out [HX_RP_SIZE][num_envs] (field-major).
 * Uniform fields hold torch.rand-style values in [0,1); normal fields hold N(0,1) draws.
 * Passing NULL makes the kernel draw the same fields from Philox4x32-10 keyed by (seed, env, step). */
#define HX_RP_DELAY 0          /* hector_env.py:166    U   x1  */
#define HX_RP_ACT_NOISE 1      /* hector_env.py:168    N   x10 */
#define HX_RP_CMD_A 11         /* legged_robot.py:308-309 resample at ep_len % 800 == 0   U x3 (x, y, heading) */
#define HX_RP_PUSH 14          /* hector_env.py:58-63  U   x5 (lin xy, ang xyz) */
#define HX_RP_RESET_Q 19       /* legged_robot.py:366  U   x10 */
#define HX_RP_RESET_XY 29      /* legged_robot.py:384  U   x2 (custom origins only) */
#define HX_RP_CMD_B 31         /* legged_robot.py:186 resample inside reset_idx        U x3 */
#define HX_RP_OBS_NOISE 34     /* hector_env.py:243    N   x41 */
#define HX_RP_SIZE 75
#define HX_RP_LEVEL 75         /* legged_robot.py:415-416 randint(max_terrain_level) as U x1.  Read ONLY when a terrain
                                  curriculum is set (hx_sim_set_terrain_curriculum): packs then have HX_RP_SIZE + 1 rows */

/* indices into hx_sim_cfg.reward_scale (already multiplied by dt, legged_robot.py:527) */
enum {
  HX_R_ACTION_SMOOTHNESS = 0, HX_R_BASE_ACC, HX_R_BASE_HEIGHT, HX_R_COLLISION, HX_R_DEFAULT_JOINT_POS,
  HX_R_DOF_ACC, HX_R_DOF_VEL, HX_R_FEET_AIR_TIME, HX_R_FEET_CLEARANCE, HX_R_FEET_CONTACT_FORCES,
  HX_R_FEET_CONTACT_NUMBER, HX_R_FEET_DISTANCE, HX_R_FOOT_SLIP, HX_R_JOINT_POS, HX_R_KNEE_DISTANCE,
  HX_R_LOW_SPEED, HX_R_ORIENTATION, HX_R_TORQUES, HX_R_TRACK_VEL_HARD, HX_R_TRACKING_ANG_VEL,
  HX_R_TRACKING_LIN_VEL, HX_R_VEL_MISMATCH_EXP
};

typedef struct hx_sim_cfg {
  int32_t num_envs;
  int32_t decimation;            /* control.decimation = 10 */
  float sim_dt;                  /* sim.dt = 1e-3 */
  float gravity_z;               /* sim.gravity[2] = -9.81 */
  /* control */
  float action_scale;            /* 0.25 */
  float clip_actions;            /* 100 */
  float clip_observations;       /* 100 */
  float default_dof_pos[HX_MAX_DOF]; /* first num_dof entries are used */
  float p_gains[HX_MAX_DOF];
  float d_gains[HX_MAX_DOF];
  float torque_limits[HX_MAX_DOF];   /* effort * safety.torque_limit */
  /* domain randomisation / noise */
  float action_delay;            /* 0.0 */
  float action_noise;            /* 0.02 */
  int32_t add_noise;
  float noise_level;             /* 0.6 */
  float noise_scale_vec[HX_MAX_OBS_FRAME];   /* first 11 + 3 * num_dof entries */
  int32_t push_robots;
  int32_t push_interval;         /* ceil(push_interval_s / dt) = 400 */
  float max_push_vel_xy;         /* 0.3 */
  float max_push_ang_vel;        /* 0.4 */
  /* commands */
  int32_t resample_interval;     /* int(resampling_time / dt) = 800 */
  int32_t heading_command;
  float cmd_range[4][2];         /* lin_vel_x, lin_vel_y, ang_vel_yaw, heading */
  /* observation scales */
  float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_quat;
  /* episode */
  float max_episode_length;      /* ceil(episode_length_s / dt) = 2400 */
  float max_episode_length_s;    /* 24 */
  float env_dt;                  /* decimation * sim_dt = 0.01 */
  float base_init_state[13];     /* pos, quat xyzw, lin vel, ang vel */
  int32_t custom_origins;        /* 1 for heightfield/trimesh (adds U[-1,1] xy on reset) */
  /* rewards */
  float reward_scale[HX_NUM_REWARDS];
  int32_t only_positive_rewards;
  float base_height_target, min_dist, max_dist, target_joint_pos_scale, target_feet_height;
  float cycle_time, tracking_sigma, max_contact_force;
  /* physics model (DESIGN.md "Physics model"; no counterpart in the reference, PhysX is opaque) */
  float contact_kn, contact_dn, friction_veps, limit_k, limit_d, terrain_mu;
  /* this object simulates envs [env_id_offset, env_id_offset + num_envs) of a larger logical batch: only the
   * random streams depend on it (Philox is keyed by the global env id) */
  int32_t env_id_offset;
  /* 10 (or 0) = hector; 18 = hector_full: observation frames 65 / 94 wide, buffers [N][976] / [N][1412], q / qd / action /
   * torque vectors of 18 in Isaac Gym's DoF order (L leg, L arm, R leg, R arm), 19 bodies in HX_BUF_CONTACT, and one more
   * reward ingredient (the arm term of default_joint_pos, hector_w_arm_env.py:371-378).  Random packs then follow the same
   * field order with 18-wide action-noise / reset rows and a 65-wide observation-noise block. */
  int32_t num_dof;
  /* sim.physx of the reference's config (hector_config.py:108-120), as far as a penalty contact model has a place for them:
   * max_depenetration_velocity (1.0) caps the speed at which a penetrating point is pushed out, contact_offset (0.01) is the
   * distance at which a point becomes a contact (its approach speed beyond gap / dt is damped before it touches),
   * rest_offset (0) the distance at which shapes rest.  0 / 0 / 0 = the plain spring-damper.  solver_type (TGS),
   * num_position_iterations, num_velocity_iterations and bounce_threshold_velocity (restitution is 0) have no counterpart:
   * nothing here iterates.  DESIGN.md 4. */
  float max_depenetration_velocity, contact_offset, rest_offset;
  /* 1: the robot's links collide with each other (asset.self_collisions = 0 in the reference's bit-filter convention,
   * humanoid_config.py:66).  Built for humanoid_ppo: knee-knee and foot-foot sphere pairs (isaac_amd/csrc/hx_dyn.h ModelXBot);
   * the hector configs switch self-collision off (hector_config.py:37) and their models carry no pairs. */
  int32_t self_collisions;
} hx_sim_cfg;

typedef struct hx_sim hx_sim;

enum hx_sim_buffer_id {
  HX_BUF_OBS = 0,          /* float [N][HX_OBS_LD], clipped, current step */
  HX_BUF_PRIV,             /* float [N][HX_PRIV_LD] */
  HX_BUF_REW,              /* float [N] */
  HX_BUF_RESET,            /* uint8 [N]  reset_buf */
  HX_BUF_TIMEOUT,          /* uint8 [N]  time_out_buf of this step */
  HX_BUF_TIMEOUT_VISIBLE,  /* uint8 [N]  extras["time_outs"]: refreshed only on steps where some env reset */
  HX_BUF_EP_LEN,           /* int32 [N]  episode_length_buf */
  HX_BUF_COMMANDS,         /* float [4][N] */
  HX_BUF_TORQUES,          /* float [10][N] torques of the last substep */
  HX_BUF_CONTACT,          /* float [11*3][N] net contact force per body, world frame */
  HX_BUF_BODY_STATE,       /* float [4*13][N] rigid_body_state of L_calf, L_toe, R_calf, R_toe */
  HX_BUF_EPISODE_SUMS,     /* float [HX_NUM_REWARDS][N] */
  HX_BUF_FEET_AIR_TIME,    /* float [2][N] */
  HX_BUF_FEET_HEIGHT,      /* float [2][N] */
  HX_BUF_NUM_RESET         /* int32 [1] number of envs that reset in the last step */
};

const char* hx_last_error(void);
int hx_version(void);

int hx_sim_create(const hx_sim_cfg* cfg, const float* shape_friction_h, const float* base_mass_h,
                  const float* env_origins_h /*[N][3]*/, const float* start_pos_h /*[N][3]*/,
                  uint64_t seed, void* hip_stream /*NULL: library creates one*/, hx_sim** out);
/* Rough terrain -- replaces gym.add_heightfield / gym.add_triangle_mesh (reference legged_robot.py:553-585, called
 * from hector_env.py:120-127) for the grid a `HumanoidTerrain` lays out (humanoid/utils/terrain.py:189-234).
 * heights_h: host int16 [rows][cols] row-major in units of vertical_scale (Terrain.heightsamples); node (i, j) lies
 * at world (x0 + i*horizontal_scale, y0 + j*horizontal_scale) -- the reference passes x0 = y0 = -border_size.
 * Collision surface: two triangles per cell split along (i,j)-(i+1,j+1), as convert_heightfield_to_trimesh emits
 * them; outside the grid the border continues.  wall_height > 0 (mesh_type 'trimesh': slope_treshold * horizontal_scale,
 * utils/terrain.py:70-73 with legged_robot_config.py:67) reproduces the slope-threshold vertex shift of that function: where
 * grid neighbours differ by more than wall_height the low ground continues flat up to the high vertices' grid line and a
 * vertical wall stands there, which collides sideways; 0 (mesh_type 'heightfield') keeps the ramps.  Call before
 * hx_sim_reset_all; heights_h == NULL returns to the ground plane z = 0 (gym.add_ground, legged_robot.py:541-551). */
int hx_sim_set_terrain(hx_sim* s, const int16_t* heights_h, int32_t rows, int32_t cols, float horizontal_scale,
                       float vertical_scale, float x0, float y0, float wall_height);
/* ablation switches for the trimesh walls (tools/falls_by_tile.py): flags & 1 = cliff cells keep their ramp, flags & 2 = no
 * sideways wall contact; 0 (default) = the reference's trimesh semantics */
int hx_sim_set_terrain_options(hx_sim* s, int32_t flags);
/* Terrain curriculum, LeggedRobot._update_terrain_curriculum (legged_robot.py:399-419) with the tables of
 * _get_env_origins (legged_robot.py:687-697).  origins_h [rows][cols][3] = terrain.env_origins, levels_h / types_h [N] =
 * terrain_levels / terrain_types, env_length = terrain.env_length, max_episode_length_s as in the config.  Every reset
 * after the constructor's then moves the robot one row up (walked > env_length / 2), one row down (walked less than
 * half the commanded distance) or to a random row (past the last), and re-bases its origin.  origins_h == NULL: off. */
int hx_sim_set_terrain_curriculum(hx_sim* s, const float* origins_h, int32_t rows, int32_t cols, const int32_t* levels_h,
                                  const int32_t* types_h, float env_length, float max_episode_length_s);
int hx_sim_get_terrain_levels(hx_sim* s, int32_t* levels_h /*[N]*/);
void hx_sim_destroy(hx_sim* s);
int hx_sim_reset_all(hx_sim* s, const float* pack /*nullable*/);
int hx_sim_step(hx_sim* s, const float* actions /*[N][10] row-major*/, const float* pack /*nullable*/);
/* hx_sim_step with zero-copy hand-over: the new observation rows ([N][HX_OBS_LD] / [N][HX_PRIV_LD]) and, if rew_dst is
 * not NULL, reward / done / extras["time_outs"] of the step are written straight into the caller's buffers (the
 * learner's rollout storage, what RolloutStorage.add_transitions copies in the reference, rollout_storage.py:87-100).
 * HX_BUF_OBS / HX_BUF_PRIV then point at obs_dst / priv_dst until the next step.  obs_dst / priv_dst must be 16-byte aligned (rows are
 * HX_OBS_LD / HX_PRIV_LD floats, multiples of 4, written with 16-byte stores) and must not be the buffers the previous step wrote
 * (the new rows are built from those, one frame down); both are checked and refused with a message. */
int hx_sim_step_ex(hx_sim* s, const float* actions, const float* pack, float* obs_dst, float* priv_dst,
                   float* rew_dst, uint8_t* done_dst, uint8_t* timeout_dst);
/* Scheduling hint for a consumer that runs background work beside the rollout (the learner's deferred critic): while `word` (device
 * memory, two int32) is set, the stacking launch of every hx_sim_step_ex adds 1 to word[0] when it starts and sets word[1] = 1; the
 * consumer's next launch takes the 1 back (word[1] tells it to).  Background kernels that sleep while word[0] is up then leave the
 * memory system to the stacking launch.  NULL (default): off.  Changes scheduling only. */
int hx_sim_set_pause_word(hx_sim* s, int32_t* word);
/* Single-frame observation storage (the rollout fast path of hx_rollout; no reference counterpart: it changes WHERE the
 * 15-frame rows of hector_env.py:246-254 live, not what they are).  The reference re-materialises every robot's
 * [15 x 41] / [15 x 70] stack each step and stores all of them (rollout_storage.py:60-61: 1.64 GB per 60-step rollout at 4096
 * robots).  Consecutive rows of one robot share 14 of their 15 frames, so here the consumer keeps each robot's frames ONCE, in
 * time order and contiguous --  frame p of env e at  base + e * env_stride + p * width  -- and row t is the window of
 * `stack` frames that ends at the frame the step before t produced; `age` = how many frames of that window are real (1..stack):
 * the reference zeroes a robot's whole history on reset (hector_env.py:256-261), so the first stack - age frames of the row
 * read as zero.  hx_sim_step_frames is hx_sim_step_ex without the stacking launch: the new (clipped) frames, the row's age and
 * the first valid element index (kz = (stack - age) * width, what the learner's loaders compare against) go straight to the
 * consumer's slots.  Reward / done / extras["time_outs"] of the step and the episode statistics are handed over by the NEXT
 * reader of the rows (they depend on this launch's total reset count): hx_sim_take_book returns the pointers and the consumer
 * runs hx_step_book_row / _global (hx_common.h) for every robot once -- the fused rollout actor does it while staging its
 * rows -- or calls hx_sim_flush_book.  hx_sim_export_stack writes the simulator's CURRENT rows as frames 0..stack-1 of every
 * robot (start of a rollout), hx_sim_import_stack rebuilds the simulator's own row buffers (HX_BUF_OBS / HX_BUF_PRIV) from
 * the consumer's frames (end of a rollout), so the row API stays valid around a frame-mode rollout. */
typedef struct hx_frame_slot {     /* where one step's outputs go */
  float* obs; int64_t obs_env_stride;     /* frame slot of env 0 and the distance between envs, in floats */
  float* priv; int64_t priv_env_stride;
  int32_t* obs_kz; int32_t* priv_kz;      /* [N] first valid element of the row this frame completes */
} hx_frame_slot;
typedef struct hx_step_book {      /* what the stacking launch does besides the rows */
  const uint8_t* reset; const uint8_t* timeout; uint8_t* timeout_visible;
  const int32_t* num_reset; int32_t* num_reset_next;
  float* stat_sum; float* stat_last; float* stat_acc; int32_t* stat_steps;
  const float* rew; float* rew_out; uint8_t* done_out; uint8_t* timeout_out;
  int32_t n;
} hx_step_book;
int hx_sim_step_frames(hx_sim* s, const float* actions, const float* pack, const hx_frame_slot* dst,
                       float* rew_dst, uint8_t* done_dst, uint8_t* timeout_dst);
int hx_sim_take_book(hx_sim* s, hx_step_book* book /*out*/, int32_t* valid /*out: 1 if a step's bookkeeping is still owed*/);
int hx_sim_flush_book(hx_sim* s);
int hx_sim_export_stack(hx_sim* s, const hx_frame_slot* first /* slot of frame 0; kz arrays = those of row 0 */);
int hx_sim_import_stack(hx_sim* s, const hx_frame_slot* first /* slot of the row's oldest frame; kz arrays = the row's */);
int hx_sim_buffer(hx_sim* s, int which, void** dptr);
int hx_sim_get_state(hx_sim* s, float* root13_h /*[N][13]*/, float* q_h /*[N][10]*/, float* qd_h /*[N][10]*/);
int hx_sim_set_state(hx_sim* s, const float* root13_h, const float* q_h, const float* qd_h);
int hx_sim_set_episode_length(hx_sim* s, const int32_t* ep_len_h);
int hx_sim_set_step_counter(hx_sim* s, int64_t common_step_counter);
int64_t hx_sim_step_counter(hx_sim* s);      /* common_step_counter (legged_robot.py:128): env steps taken so far */
/* play-style access (reference humanoid/scripts/play.py:136-140,160-175): overwrite `env.commands` [N][4]
 * (vx, vy, yaw rate, heading) before a step, and read `env.base_lin_vel` / `env.base_ang_vel` [N][3] (base frame,
 * legged_robot.py:132-133) after it.  Host pointers; both synchronise the simulator's stream. */
int hx_sim_set_commands(hx_sim* s, const float* commands_h);
int hx_sim_get_base_velocities(hx_sim* s, float* lin_h, float* ang_h);
/* What the runner logs as Episode/rew_* and Train/mean_*: per step with a reset, extras["episode"]["rew_k"] = mean over
 * that step's resets of episode_sum_k / max_episode_length_s (legged_robot.py:198-201); the dict persists on steps without
 * a reset and the runner appends it every step, then averages over the steps of the iteration
 * (on_policy_runner.py:141-142,181-195) -> mean_h[0..HX_NUM_REWARDS).  mean_h[HX_NUM_REWARDS], [+1]: mean return and length
 * of the last <= 100 finished episodes (the rewbuffer / lenbuffer deques, on_policy_runner.py:112-113,140-154).
 * count_h: episodes finished since the last call.  The call starts a new iteration's accumulation. */
int hx_sim_episode_stats(hx_sim* s, float* mean_h /*[HX_NUM_REWARDS + 2]*/, int32_t* count_h);
void* hx_sim_stream(hx_sim* s);
/* identity of this build: "<version>-<first 16 hex digits of the SHA-256 over the library's sources>" (isaac_amd/build.py);
 * measurement files under profiles/ name the build they were taken on with it */
const char* hx_build_id(void);
int hx_sync(void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
