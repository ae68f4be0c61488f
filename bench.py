#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the hector PPO hot path on MI355X (BASELINE.json metric).

One "step" = one full training iteration of the reference loop (on_policy_runner.py:124-170) on one batch of
synthetic-but-real work: 60 x {actor/critic forward + sample, env step with 10 physics substeps, transition
store} for `--envs` (default 4096) robots per GPU, then GAE and the PPO update (2 epochs x 4 minibatches of
61 440 rows, fwd + loss + bwd + clip + Adam).  Nothing is skipped or cached; weights are random-init
(no checkpoints exist offline), robots are the real hector model, all inputs already live in HBM.

value = num_steps_per_env * num_envs * n_gpus * K / max-over-ranks(elapsed)      [env-steps / s, whole job]
       = the reference's own `Perf/total_fps` (on_policy_runner.py:199-203).

Launch: `python bench.py` (1 GPU) or
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
         bench.py --gpus N --steps K --warmup W`   (one rank per GPU, weak scaling: 4096 envs on every rank,
         one RCCL all-reduce of the flat gradient per optimiser step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_ENV_STEP = 16_772_066            # dense MLP flops, rollout 3 066 338 + update 13 705 728
PEAK_F32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak


def pmc_traffic(kernel_name):
    """HBM-side bytes per launch of `kernel_name` from the committed PMC pass (profiles/*_traffic.json, written by
    tools/collect_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same
    command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The counters cannot be read live
    from inside the timed run, so the newest committed file for the default workload is reported; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d["kernels"][kernel_name]
        return k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"], {
            "unit": "bytes per launch (fetch + write)", "fetch_bytes": k["fetch_bytes_per_launch"], "write_bytes": k["write_bytes_per_launch"],
            "launches_sampled": k["launches_sampled"], "source": "profiles/" + os.path.basename(files[-1])}
    except Exception:
        return None, None


def cpu_baseline(sample_envs=4096, sample_steps=6, terrain="trimesh"):
    """The numpy oracle (oracle/env.py + oracle/physics.py + oracle/ppo.py: the CPU restatement, kind="port")
    timed on this box's host cores for a bounded sample of the same iteration: `sample_steps` rollout steps of
    `sample_envs` robots (policy forward, env step with 10 substeps, store), GAE and the 2x4-minibatch update."""
    from oracle.env import HectorEnvOracle, RP_SIZE
    from oracle.ppo import ActorCriticOracle, PPOOracle
    rng = np.random.default_rng(0)
    n, T = sample_envs, sample_steps
    pack = lambda: np.concatenate([rng.uniform(size=(34, n)), rng.standard_normal((41, n))]).astype(np.float32)
    hf, origins = None, np.zeros((n, 3))
    if terrain != "plane":                # the same 20 x 20 tile map layout as the product's default config (untimed set-up)
        import types
        from oracle.terrain import HeightField, HumanoidTerrainOracle
        tc = types.SimpleNamespace(mesh_type=terrain, horizontal_scale=0.1, vertical_scale=0.005, border_size=25, curriculum=False,
                                   selected=False, terrain_length=8.0, terrain_width=8.0, num_rows=20, num_cols=20,
                                   terrain_proportions=[0.1, 0.1, 0.2, 0.1, 0.1, 0.2, 0.2])
        np.random.seed(5)
        ter = HumanoidTerrainOracle(tc, n)
        hf = HeightField(ter.heightsamples, 0.1, 0.005, 25)
        origins = ter.env_origins[rng.integers(0, 20, n), np.floor(np.arange(n) / (n / 20)).astype(int)]
    env = HectorEnvOracle(n, rng.uniform(0.1, 1.0, n), 8.15528 + rng.uniform(-2, 4, n), origins, pack(),
                          start_xy=origins, terrain=hf, custom_origins=hf is not None)
    alg = PPOOracle(ActorCriticOracle.default_init(rng), n, T)
    t0 = time.perf_counter()
    obs, priv = env.obs_buf, env.priv_buf
    for _ in range(T):
        a = alg.act(obs, priv, rng.standard_normal((n, 10)).astype(np.float32))
        obs, priv, rew, done = env.step(a, pack())
        alg.process_env_step(rew, done, env.time_outs_visible)
    alg.compute_returns(priv)
    alg.update(rng.permutation(n * T))
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": n * T / dt, "unit": "env-steps/s", "cores": int(threads), "kind": "port",
            "sample": f"{T} rollout steps x {n} envs (terrain {terrain}) + GAE + full 2x4-minibatch update, numpy float32/64 oracle, "
                      f"{dt:.1f} s wall; host has {os.cpu_count()} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--task", default="hector", choices=["hector", "hector_full"],
                    help="hector is BASELINE.json's metric config; hector_full (18 DoF, SURVEY 8f-4) is a side measurement")
    ap.add_argument("--shards", type=int, default=1, help="env shards per GPU driven round-robin on separate streams (1 = off; measured slower than the deferred-critic overlap, see DESIGN.md)")
    ap.add_argument("--terrain", default="trimesh", choices=["trimesh", "heightfield", "plane"],
                    help="terrain.mesh_type; 'trimesh' is the reference's default for the hector task (hector_config.py:45)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="learner precision: f32 = the metric's configuration (BASELINE config 2); bf16 = BASELINE config 4 "
                         "(forward/dgrad products on the bf16 matrix cores, fp32 master weights and wgrads) -- a different "
                         "configuration, reported with dtype 'bf16' and never as the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic, one rank only: walk the N > 1 code path (RCCL all-reduce per optimiser step, as identity) "
                         "to see what the distributed orchestration costs before any link time")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    args = ap.parse_args()

    import contextlib
    # stdout carries exactly one line: the result JSON.  Python-level prints go to stderr, and so does file descriptor 1
    # itself while the job runs: RCCL prints a version banner to fd 1 from native code when its first communicator comes up.
    sys.stdout.flush()
    real_fd = os.dup(1)
    os.dup2(2, 1)
    real_stdout = os.fdopen(real_fd, "w")
    sys.stdout = sys.stderr
    import __graft_entry__
    __graft_entry__.build()
    from isaac_amd import capi
    from isaac_amd.parallel import init_comm
    from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO, HectorFullCfg, HectorFullCfgPPO
    from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv, PipelinedHectorEnv, class_to_dict
    from isaac_amd.algo.on_policy_runner import OnPolicyRunner
    from isaac_amd.utils.helpers import set_seed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    if args.force_collectives and world == 1:
        from isaac_amd.parallel import TorchComm
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        comm = TorchComm("nccl")
        comm.force_collectives = True
    else:
        comm = init_comm()
    ndev = max(1, capi.lib().hx_device_count())
    local = comm.local_rank % ndev       # one rank per GPU under the driver; wraps only in the one-GPU gloo-staged rehearsal
    capi.check(capi.lib().hx_set_device(local), "hx_set_device")

    full = args.task == "hector_full"
    if full and args.shards > 1:
        raise SystemExit("--shards applies to the hector task only")
    env_cfg, train_cfg = (HectorFullCfg(), HectorFullCfgPPO()) if full else (HectorCfg(), HectorCfgPPO())
    env_cfg.env.num_envs = args.envs
    env_cfg.terrain.mesh_type = args.terrain
    env_cfg.seed = set_seed(train_cfg.seed + comm.rank)
    if args.shards > 1:
        env = PipelinedHectorEnv(env_cfg, sim_device=f"cuda:{local}", headless=True, num_shards=args.shards)
    else:
        env = (HectorFullFreeEnv if full else HectorFreeEnv)(env_cfg, sim_device=f"cuda:{local}", headless=True)
    tcfg = class_to_dict(train_cfg)
    if args.dtype != "f32":
        tcfg["algorithm"]["mlp_dtype"] = args.dtype          # extra PPO keyword of this build (hx_ppo_set_compute_dtype)
    runner = OnPolicyRunner(env, tcfg, log_dir=None, device=f"cuda:{local}", comm=comm)
    T = runner.num_steps_per_env

    # HIP-event brackets cost GPU time (every GEMM launch bracketed: -1.5 % env-steps/s), so the full per-kernel table is
    # taken during the untimed warm-up iterations and only the dominant kernel found there is bracketed in the timed region.
    prof_all = None
    if not args.no_prof and args.warmup > 0:
        runner.alg.prof_begin()
    runner.learn(args.warmup, init_at_random_ep_len=True)          # untimed warm-up iterations
    env.sync()
    if not args.no_prof and args.warmup > 0:
        prof_all = runner.alg.prof_end()
    comm.barrier()
    if not args.no_prof:
        dominant = max(prof_all["kernels"], key=lambda r: r["ms"])["name"] if prof_all and prof_all["kernels"] else None
        runner.alg.prof_begin(only=dominant)
    t0 = time.perf_counter()
    runner.learn(args.steps, init_at_random_ep_len=False)
    env.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    prof = None if args.no_prof else runner.alg.prof_end()
    elapsed = comm.max_over_ranks(elapsed)

    if comm.rank == 0:
        env_steps = T * args.envs * world * args.steps
        value = env_steps / elapsed
        dims = lambda d: "[" + ",".join(str(x) for x in d) + "]"
        out = {"metric": f"env-steps/sec (whole node), {args.task} {args.envs} envs/GPU", "value": value, "unit": "env-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{args.task} {args.envs} envs/GPU, 1 iteration = 60 env steps (10 x 1 ms substeps) + PPO "
                                      f"update {train_cfg.algorithm.num_learning_epochs} epochs x {train_cfg.algorithm.num_mini_batches} minibatches, fp32 HIP sim + MLP actor {dims(train_cfg.policy.actor_hidden_dims)} / "
                                      f"critic {dims(train_cfg.policy.critic_hidden_dims)}"
                                      + ("" if args.dtype == "f32" else " (bf16 forward/dgrad MFMA, fp32 master weights)"),
                          "num_envs_per_gpu": args.envs, "num_steps_per_env": T, "parallelism": f"dp{world}",
                          "terrain": args.terrain, "env_shards": args.shards, **({"forced_collectives": True} if args.force_collectives else {}), "collection_s": runner.last_perf.get("collection_time"),
                          "learn_s": runner.last_perf.get("learn_time")}}
        if prof is not None and prof["kernels"]:
            k = max(prof["kernels"], key=lambda r: r["ms"])
            achieved = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            traffic, traffic_detail = pmc_traffic(k["name"]) if (args.envs == 4096 and args.terrain == "trimesh" and args.shards == 1 and not full) else (None, None)
            # bf16 mode: the same kernel ids run on the bf16 matrix cores (dense peak 2.5 PFLOP/s); with fp32 operands in HBM
            # those products are memory-bound, which is what the small fraction of that peak says
            peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else 2500.0
            out["roofline"] = {"bound": "mfma", "kernel": k["name"] + ("" if args.dtype == "f32" else " [bf16 operands]"), "achieved": achieved, "peak": peak,
                               "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic if args.dtype == "f32" else None,
                               "traffic_detail": traffic_detail,
                               "launches": k["launches"], "avg_launch_us": 1e3 * k["ms"] / max(1, k["launches"]),
                               "flop_per_launch": k["flops"] / max(1, k["launches"]),
                               "all_gemm_kernels": (prof_all or prof)["kernels"],
                               "all_gemm_kernels_from": "warm-up iterations (all launches bracketed)" if prof_all else "timed region",
                               "whole_iteration_mfma_frac": None if full else value / world * FLOP_PER_ENV_STEP / (peak * 1e12)}
        if not args.no_cpu_baseline and world == 1 and not full:            # rank 0 at N = 1 only; other ranks wait at the barrier below
            try:
                out["cpu_baseline"] = cpu_baseline(terrain=args.terrain)
            except Exception as e:                        # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), file=real_stdout, flush=True)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
