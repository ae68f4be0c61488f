/* hx_lab.h -- measurement and unit-test hooks of libhx.so.  NOT part of the drop-in boundary: a maintainer binding the
 * library for the reference's VecEnv / PPO seams needs include/hx_sim.h and include/hx_ppo.h only (INTEGRATION.md cites
 * nothing from this file).  These entry points exist so that every number under profiles/ can be reproduced and so that
 * the GEMM kernels can be tested in isolation; they change no result of the product path. */
#ifndef HX_LAB_H
#define HX_LAB_H
#include <stdint.h>
#include "hx_ppo.h"
#include "hx_sim.h"
#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event timing of the learner's GEMM launches on the learner's stream, ONE row per kernel symbol, named exactly as
 * rocprofv3 prints it (e.g. "hx_gemm_group_kernel<128, 128, 16, false, false, 2, true>"), so a row here and a row of
 * `rocprofv3 --kernel-trace --stats` are the same launches.  hx_ppo_prof_begin(p, NULL, 1) brackets every symbol;
 * (p, symbol, n) only that one, every n-th of its launches.  An event pair is not free: the kernel trace shows ~7 us of idle stream on either side of a
 * bracketed launch and none between unbracketed ones (profiles/r03_p_grouped_launches.txt), 0.6 ms per iteration when every launch of the dominant
 * symbol carries one -- so a benchmark brackets a uniform SAMPLE of the launches in its timed region (sample_every).  The deferred critic's launches
 * on the background stream are never bracketed (they overlap the rollout's kernels).  hx_ppo_prof_end stops and returns
 * the rows with at least one launch. */
typedef struct hx_prof_row { char symbol[128]; double ms; int64_t launches; double flops; } hx_prof_row;
int hx_ppo_prof_begin(hx_ppo* p, const char* only_symbol /*nullable*/, int sample_every /*<= 1: every launch; n: a uniform sample, every n-th launch of the selected symbols*/);
int hx_ppo_prof_end(hx_ppo* p, hx_prof_row* rows_h, int max_rows, int* n_rows);

/* unit-test hook: one GEMM of the given mode (0/3 fwd bias+ELU 128/64-row tile, 1/4 dgrad * elu', 2 wgrad single
 * split; add 10 for the BK = 32 variant; 5-9: bf16 and persistent-grid variants, tests/test_gpu_gemm.py) */
int hx_ppo_gemm_test(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     const float* bias, float* C, int ldc, const float* H, void* hip_stream);
/* unit-test hook: the weight gradients dW_l[out_l][in_ld_l] = dZ_l[rows][out_l]^T X_l[rows][in_ld_l] and bias gradients (column sums of
 * dZ_l) of up to six layers through the one-workgroup-per-CU path of the update (planner hx_wgrad_plan.h, hx_wgrad_multi_kernel, slab
 * reduction).  The pointer tables are HOST arrays of DEVICE pointers; slots <= 0: one workgroup per CU. */
int hx_ppo_wgrad_multi_test(int nl, const int* out_h, const int* in_ld_h, int rows, const float* const* dZ_h, const float* const* X_h,
                            float* const* dW_h, float* const* db_h, int slots, int* nlaunch_h, void* hip_stream);
/* the plan behind it, computed on the host only (no GPU call): 12 numbers per piece -- layer, first column, columns, shape id, tile rows, tile
 * columns, tiles, slices, rows per slice, launch, slab offset (floats), bias-slab offset or -1 */
int hx_wgrad_plan_describe(int nl, const int* out_h, const int* in_ld_h, int rows, int slots, long long* pieces_h, int max_pieces, int* n_pieces,
                           int* nlaunch_h, long long* slab_floats_h);
/* timing hook: mean ms per launch of one learner GEMM (kind 0 fwd, 1 dgrad, 2 wgrad split-K; bk 16 or 32) */
int hx_ppo_gemm_bench(int kind, int bk, int rows, int out, int in_ld, int iters, float* ms_out);
/* measurement hook: TFLOP/s of 256-thread workgroups whose waves issue n v_mfma_f32_32x32x2_f32 each with, by mode,
 * 0 nothing else, 1 + the GEMM's LDS fragment reads, 2 + a barrier per BK=16 tile, 3 + the tile's global loads and LDS
 * stores -- where between the matrix-pipe peak and the GEMM kernels the throughput goes (tools/mfma_peak.py) */
int hx_mfma_probe(int mode, int blocks, int n, float* tflops_out);

/* HIP-event time of the env-step kernel launches on the simulator's stream; which = 1 start / clear, 0 stop and read
 * {total milliseconds, launches} (bench.py: the live duration behind roofline.env_step) */
int hx_sim_time(hx_sim* s, int which, double* out_h /*[2]*/);
/* per-phase shader-clock cycles of the env-step kernel, summed over waves and launches; meaningful only in a library
 * built with -DHX_STEP_PROF (tools/step_prof.py), all zeros otherwise.  which = 1 start / clear, 0 read. */
int hx_sim_prof(hx_sim* s, int which, long long* out_h /*[18]: 9 phase timers, shape-visit counts [9..14], unused, then the sums over waves of the shader-clock
                                                           and of the 100 MHz wall-clock ticks a wave lived: their ratio x 0.1 = the kernel's clock in GHz*/);

/* phase stamps of the fused rollout actor's last launch, [blocks][8] 100 MHz ticks (start, rows staged, after layer 1, 2, 3, head, log-prob);
 * only in a library built with HX_EXTRA_FLAGS_HX_PPO="-DHX_ACTOR_PROF" (tools/actor_prof.py) */
int hx_ppo_actor_stamps(hx_ppo* p, long long* out_h, int blocks);
/* the learner's two pause words (hx_sim_set_pause_word): 0, 0 whenever no rollout launch is in flight */
int hx_ppo_pause_words(hx_ppo* p, int32_t* out_h);
/* lifetimes of the first n env-step waves (wave w = robots 8w .. 8w+7), 100 MHz ticks summed over the launches since hx_sim_prof(s, 1, ..);
 * -DHX_STEP_PROF builds (tools/env_clock.py: which waves make a launch as long as it is) */
int hx_sim_prof_waves(hx_sim* s, long long* out_h, int n);
/* the most recent env-step launch, first n waves: out_h[18][n] = start, end (100 MHz device-wide ticks), (XCC_ID << 32) | HW_ID, the
 * wave's shader-clock cycles in the nine phases hx_sim_prof reports, the shapes its contact loop visited (all substeps), the visits
 * of the first five shapes */
int hx_sim_prof_last(hx_sim* s, long long* out_h, int n);

#ifdef __cplusplus
}
#endif
#endif
