/* hx_ppo.h -- C ABI of the MI355X-native PPO learner (rsl_rl-style ActorCritic + PPO + RolloutStorage).
 *
 * Reference interfaces replaced (all under humanoid/algo/ppo/):
 *   hx_ppo_create            ActorCritic.__init__ (actor_critic.py:37-83) + PPO.__init__ (ppo.py:41-80)
 *                            + PPO.init_storage / RolloutStorage.__init__ (ppo.py:82-83, rollout_storage.py:52-85)
 *   hx_ppo_set/get_params_h  ActorCritic.state_dict()/load_state_dict(): tensors in `parameters()` order
 *                            std, actor.{0,2,4,6}.{weight,bias}, critic.{0,2,4,6}.{weight,bias}
 *                            (checkpoint format of on_policy_runner.py:278-295)
 *   hx_ppo_set/get_opt_state_h  torch.optim.Adam state (exp_avg, exp_avg_sq, step) in the same order
 *   hx_ppo_act               PPO.act (ppo.py:91-101): actor forward, sample, log-prob, critic forward, stash transition
 *   hx_ppo_process_step      PPO.process_env_step (ppo.py:103-113) + RolloutStorage.add_transitions (:87-100)
 *   hx_ppo_compute_returns   PPO.compute_returns (ppo.py:115-117) + RolloutStorage.compute_returns GAE loop (:122-132)
 *   hx_ppo_adv_normalize     RolloutStorage.compute_returns advantage normalisation (:135-136)
 *   hx_ppo_update            PPO.update (ppo.py:119-184) + RolloutStorage.mini_batch_generator (:146-182)
 *   hx_ppo_update_begin / _minibatch_backward / _minibatch_step / _update_end
 *                            the same loop split at the point where data-parallel ranks exchange gradients
 *                            (absent in the reference): backward fills one flat buffer
 *                            [grads..., kl_sum, value_loss_sum, surrogate_loss_sum, rows] that the host
 *                            all-reduces (RCCL) before _minibatch_step applies LR schedule, clip and Adam
 *   hx_ppo_inference         ActorCritic.act_inference (actor_critic.py:122-124)
 *
 * Conventions as in hx_sim.h: DEVICE pointers unless suffixed _h; one stream; 0 = ok; no CPU path.
 */
#ifndef HX_PPO_H
#define HX_PPO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hx_ppo_cfg {
  int32_t num_envs, num_steps;            /* N, T (num_steps_per_env) */
  int32_t num_obs, num_priv, num_actions; /* 615, 1050, 10 for hector; any widths, 1 <= num_actions <= 32 */
  int32_t actor_hidden[3], critic_hidden[3]; /* multiples of 4; the last width of each network a multiple of 64 (the two
                                               may differ: hector_full has 128 / 768, hector_w_arm_config.py:213-214) */
  int32_t num_learning_epochs, num_mini_batches;
  float clip_param, gamma, lam, value_loss_coef, entropy_coef, learning_rate, max_grad_norm;
  int32_t use_clipped_value_loss;
  int32_t adaptive_schedule;              /* schedule == 'adaptive' */
  float desired_kl;
  float init_noise_std;
  int32_t obs_ld, priv_ld;                /* row strides of the obs / privileged-obs buffers given to hx_ppo_act */
  /* Single-frame observation storage (include/hx_sim.h hx_sim_step_frames; 0 / 0 / 0 / 0 = rows are stored as given, the
   * reference's layout).  With obs_frame * obs_stack == num_obs and priv_frame * priv_stack == num_priv the rollout storage
   * keeps every robot's frames once -- [N][T + stack][frame] instead of [T][N][stack * frame]: 135 MB instead of 1.64 GB at
   * 4096 robots -- and the first-layer products of both networks (forward and weight gradient), the fused rollout actor and
   * the deferred critic read their rows through (row start, first valid element) tables straight from it; no stacking
   * launch, no minibatch gather, bit-identical results.  Such a learner is driven by hx_rollout (one simulator, fp32 mode);
   * hx_ppo_act / hx_ppo_process_step, which take ready-made rows, are refused. */
  int32_t obs_frame, priv_frame, obs_stack, priv_stack;
} hx_ppo_cfg;

typedef struct hx_ppo hx_ppo;

enum hx_ppo_buffer_id {
  HX_PPO_BUF_ACTIONS = 0,   /* float [T][N][A] */
  HX_PPO_BUF_VALUES,        /* float [T][N] */
  HX_PPO_BUF_LOGP,          /* float [T][N] */
  HX_PPO_BUF_MU,            /* float [T][N][A] */
  HX_PPO_BUF_REWARDS,       /* float [T][N] (after the time-out bootstrap) */
  HX_PPO_BUF_RETURNS,       /* float [T][N] */
  HX_PPO_BUF_ADVANTAGES,    /* float [T][N] */
  HX_PPO_BUF_GRADS,         /* float [padded params + 4] flat gradient + statistics buffer */
  HX_PPO_BUF_PERM,          /* int32 [T*N] minibatch permutation in use */
  HX_PPO_BUF_OBS,           /* float [T][N][obs_ld]   RolloutStorage.observations (rollout_storage.py:60); with frame storage
                               the rows do not exist in memory: hx_ppo_storage_rows expands them on request */
  HX_PPO_BUF_PRIV,          /* float [T][N][priv_ld]  RolloutStorage.privileged_observations */
  HX_PPO_BUF_DONES,         /* uint8 [T][N] */
  HX_PPO_BUF_TIMEOUTS       /* uint8 [T][N]  infos["time_outs"] as the step saw them (stale on steps without a reset) */
};

int hx_ppo_create(const hx_ppo_cfg* cfg, void* hip_stream, void* ext_grad_buffer /*nullable*/, hx_ppo** out);
void hx_ppo_destroy(hx_ppo* p);
void* hx_ppo_stream(hx_ppo* p);
int64_t hx_ppo_num_params(hx_ppo* p);                 /* torch element count, 1 517 973 for hector */
int hx_ppo_set_params_h(hx_ppo* p, const float* flat_h);
int hx_ppo_get_params_h(hx_ppo* p, float* flat_h);
int hx_ppo_set_opt_state_h(hx_ppo* p, const float* exp_avg_h, const float* exp_avg_sq_h, int64_t step);
int hx_ppo_get_opt_state_h(hx_ppo* p, float* exp_avg_h, float* exp_avg_sq_h, int64_t* step);
/* Precision of the learner's dense products (no reference counterpart: the reference is fp32 throughout).
 *   0  fp32 on v_mfma_f32_32x32x2_f32 everywhere (default; the configuration the parity tests and bench.py use)
 *   1  BASELINE config 4, "bf16 MLP on MFMA": the forward, input-gradient and weight-gradient products of the hidden
 *      layers round their fp32 operands to bf16 on the way into LDS and run on v_mfma_f32_32x32x16_bf16 with fp32
 *      accumulation (the fused rollout actor rounds the same operands and keeps its fp32 MFMA, so rollout and update
 *      agree); master weights, Adam, bias gradients, the output layers and the loss head stay fp32.  The learner then
 *      keeps fp32 transposed copies of the hidden weights for the dgrads. */
int hx_ppo_set_compute_dtype(hx_ppo* p, int dtype);

/* Random streams of the learner (the reference draws both from torch's global generator, actor_critic.py:117 and
 * rollout_storage.py:149; here they are counter-based).  sample_seed keys the action noise: data-parallel ranks pass
 * seed + rank so their exploration is independent; perm_seed keys the minibatch permutation.  The counters are the
 * positions in the two streams (a checkpoint stores them so that a resumed run does not replay the noise). */
int hx_ppo_set_seed(hx_ppo* p, uint64_t sample_seed, uint64_t perm_seed);
int hx_ppo_get_rng_state(hx_ppo* p, uint32_t* act_counter, uint32_t* perm_counter);
int hx_ppo_set_rng_state(hx_ppo* p, uint32_t act_counter, uint32_t perm_counter);

int hx_ppo_act(hx_ppo* p, const float* obs, const float* priv, const float* eps /*[N][A], nullable*/, float** actions_out);
int hx_ppo_process_step(hx_ppo* p, const float* rewards, const uint8_t* dones, const uint8_t* time_outs /*nullable*/);
int hx_ppo_compute_returns(hx_ppo* p, const float* last_priv /*NULL: values set by hx_ppo_last_values_range*/);
/* Shard forms for a rollout that runs as several independent env shards, each on its own stream, so that one
 * shard's (latency-bound) env step overlaps another shard's GEMMs.  Rows [env0, env0+count) of the same slot. */
int hx_ppo_act_range(hx_ppo* p, const float* obs, const float* priv, const float* eps, int env0, int count,
                     void* hip_stream, float** actions_out);
int hx_ppo_process_step_range(hx_ppo* p, const float* rewards, const uint8_t* dones, const uint8_t* time_outs,
                              int env0, int count, void* hip_stream, int advance_slot);
int hx_ppo_last_values_range(hx_ppo* p, const float* last_priv, int env0, int count, void* hip_stream);
int hx_ppo_adv_moments(hx_ppo* p, void** moments /* double[3] on device: sum, sum of squares, count */);
int hx_ppo_adv_normalize(hx_ppo* p);

/* ---- data parallelism (absent in the reference): one process per GPU, environments sharded, parameters
 * replicated.  hx_comm is an RCCL communicator over xGMI owned by this library (bound at run time, no link dependency):
 *   rank 0:      hx_comm_get_unique_id(id)    -- 128 opaque bytes (ncclUniqueId), handed to the other ranks by the host
 *   every rank:  hx_comm_init(id, rank, world, &c) ; hx_ppo_set_comm(learner, c)
 * From then on hx_ppo_update / hx_ppo_minibatch_step enqueue ONE ncclAllReduce(sum) of the flat
 * [gradient | kl_sum | value_loss_sum | surrogate_sum | rows] buffer per optimiser step on the learner's stream, between the
 * backward kernels and the Adam kernel (which scales by 1 / world), and hx_ppo_adv_normalize all-reduces the three
 * advantage moments on the same stream first, so that the normalisation (rollout_storage.py:135-136) is the one a single
 * process would compute over all ranks' rows.  Nothing synchronises with the host.  Every rank sees the same global KL, so
 * the learning-rate decision (ppo.py:136-148) and with it the parameters stay bit-identical without a broadcast;
 * hx_ppo_broadcast_params makes them identical once at start-up.
 * Binding and failure behaviour: librccl.so.1 is resolved at run time -- the copy the process already maps (PyTorch's) if
 * there is one (dlopen RTLD_NOLOAD), else the system's -- and must report the NCCL API major version this library was built
 * against (ncclGetVersion; hx_comm_library_info).  No wait is unbounded: hx_comm_init returns an error if ncclCommInitRank
 * has not come back after HX_COMM_INIT_TIMEOUT_S (300 s), and every host synchronisation with a stream that carries
 * collectives goes through hx_comm_wait, which after HX_COMM_TIMEOUT_S (120 s) or on an asynchronous RCCL error aborts the
 * communicator (ncclCommAbort) and returns an error naming the rank -- the process is expected to exit non-zero then; the
 * launcher may start a fresh one (a process that has touched the GPU is never re-exec'd). */
#define HX_COMM_ID_BYTES 128
enum { HX_COMM_F32 = 0, HX_COMM_F64 = 1 };
enum { HX_COMM_SUM = 0, HX_COMM_MAX = 1 };
typedef struct hx_comm hx_comm;
int hx_comm_get_unique_id(uint8_t* id_h /*[HX_COMM_ID_BYTES]*/);
int hx_comm_init(const uint8_t* id_h, int rank, int world, hx_comm** out);
void hx_comm_destroy(hx_comm* c);
int hx_comm_rank(hx_comm* c);
int hx_comm_world(hx_comm* c);
/* in-place collectives on device buffers for the host's bookkeeping (barrier, max-over-ranks timing, episode statistics) */
int hx_comm_all_reduce(hx_comm* c, void* buf, size_t count, int dtype, int op, void* hip_stream);
int hx_comm_broadcast(hx_comm* c, void* buf, size_t count_f32, int root, void* hip_stream);
/* wait for everything enqueued on hip_stream with a deadline (timeout_s <= 0: the communicator's HX_COMM_TIMEOUT_S);
 * c == NULL: plain stream synchronisation */
int hx_comm_wait(hx_comm* c, void* hip_stream, double timeout_s);
int hx_comm_library_info(int* nccl_version, int* shared_copy /*1: the process had librccl.so.1 mapped already*/);
/* ranks that split ONE logical batch pass the global index of their first env row: the action-noise stream is keyed by the
 * global row then, so the job samples what a single process on the union of the shards would (tests/test_gpu_dp.py) */
int hx_ppo_set_row_base(hx_ppo* p, uint32_t first_global_row);
int hx_ppo_set_comm(hx_ppo* p, hx_comm* c /*NULL: back to single process*/);
int hx_ppo_broadcast_params(hx_ppo* p, int root);

int hx_ppo_update_begin(hx_ppo* p, const int32_t* perm /*[T*N], nullable: drawn on device*/);
int hx_ppo_minibatch_backward(hx_ppo* p, int mb_index, void** grad_buffer, int64_t* count);
int hx_ppo_minibatch_step(hx_ppo* p, float inv_world_size);
int hx_ppo_update_end(hx_ppo* p, float* stats_h /*[4]: mean value loss, mean surrogate loss, lr, last kl*/);
int hx_ppo_update(hx_ppo* p, const int32_t* perm, float* stats_h);

/* `steps` iterations of {PPO.act, env.step, PPO.process_env_step} (on_policy_runner.py:127-138) without returning
 * to the host language; sims[h] simulates env rows [env0[h], env0[h]+count[h]).  hx_sim is declared in hx_sim.h. */
struct hx_sim;
int hx_rollout(hx_ppo* p, struct hx_sim** sims, const int32_t* env0, const int32_t* count, int nshards, int steps);
/* RolloutStorage.observations / .privileged_observations of slots [t0, t1) as rows [t1 - t0][N][ld], whatever the storage
 * layout (frame storage: expanded from the frames with the zeroed history of reset robots; t1 may be num_steps + 1 to
 * include the rows after the last step).  which = HX_PPO_BUF_OBS or HX_PPO_BUF_PRIV; dst = device memory of the caller. */
int hx_ppo_storage_rows(hx_ppo* p, int which, int t0, int t1, float* dst);

int hx_ppo_buffer(hx_ppo* p, int which, void** dptr);
int hx_ppo_get_lr(hx_ppo* p, float* lr_h);
int hx_ppo_set_lr(hx_ppo* p, float lr);
int hx_ppo_inference(hx_ppo* p, const float* obs, int rows, float* actions_out);
/* device memory helpers for hosts without a tensor library */
int hx_malloc(size_t bytes, void** out);
int hx_free(void* ptr);
int hx_memcpy_h2d(void* dst, const void* src_h, size_t bytes, void* hip_stream);
int hx_memcpy_d2h(void* dst_h, const void* src, size_t bytes, void* hip_stream);
int hx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* hip_stream);
int hx_device_count(void);
int hx_set_device(int index);

#ifdef __cplusplus
}
#endif
#endif
