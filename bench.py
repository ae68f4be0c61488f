#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the hector PPO hot path on MI355X (BASELINE.json metric).

One "step" = one full training iteration of the reference loop (on_policy_runner.py:124-170) on one batch of
synthetic-but-real work: 60 x {actor/critic forward + sample, env step with 10 physics substeps, transition
store} for `--envs` (default 4096) robots per GPU, then GAE and the PPO update (2 epochs x 4 minibatches of
61 440 rows, fwd + loss + bwd + clip + Adam).  Nothing is skipped or cached; weights are random-init
(no checkpoints exist offline), robots are the real hector model, all inputs already live in HBM.

value = num_steps_per_env * num_envs * n_gpus * K / max-over-ranks(elapsed)      [env-steps / s, whole job]
       = the reference's own `Perf/total_fps` (on_policy_runner.py:199-203).

Launch: `python bench.py` (1 GPU), or for N GPUs of one node either
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
         bench.py --gpus N --steps K --warmup W`   (the driver's form), or simply
        `python bench.py --gpus N --steps K --warmup W`: without WORLD_SIZE in the environment this process starts the N
         ranks itself as child processes (before it touches the GPU or the library; never by exec) and relays rank 0's line.
        One rank per GPU, weak scaling: 4096 envs on every rank, one RCCL all-reduce of the flat gradient per optimiser step.
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


def _committed(pattern, build_id):
    """Newest committed measurement file under profiles/ matching `pattern` that was taken ON THIS BUILD: the file's
    "build_id" (written by tools/collect_*.py from hx_build_id()) must equal the running library's.  A file from another
    build describes other kernels -- it is refused (None), never reported beside this run's timings."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get("build_id") == build_id:
            d["source"] = "profiles/" + os.path.basename(path)
            return d
    return None


def pmc_traffic(kernel_symbol, build_id):
    """HBM-side bytes per launch of `kernel_symbol` from the committed PMC pass (profiles/*_traffic.json, written by
    tools/collect_traffic.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The counters cannot be read live from inside the
    timed run; None unless a file of this very build (hx_build_id) holds this very symbol."""
    d = _committed("*_traffic.json", build_id)
    k = (d or {}).get("kernels", {}).get(kernel_symbol)
    if not k:
        return None, None
    return k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"], {
        "unit": "bytes per launch (fetch + write)", "fetch_bytes": k["fetch_bytes_per_launch"], "write_bytes": k["write_bytes_per_launch"],
        "launches_sampled": k["launches_sampled"], "source": d["source"], "build_id": d["build_id"]}


def pmc_env_step(build_id):
    """Wave-level VALU instructions per launch of the env-step kernel from the committed PMC pass of this build
    (profiles/*_env_step_pmc.json, `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES` on this command); None otherwise."""
    return _committed("*_env_step_pmc.json", build_id)


def host_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box hands a job a
    share of its host, e.g. 16 of 256 logical cores: an OpenMP team sized by os.cpu_count() would spend its time in barriers
    waiting for descheduled threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(sample_envs=4096, sample_steps=60, terrain="trimesh", update_rows=61440):
    """CPU baseline on this box's host cores, kind = "port" (SURVEY.md 8d ii: the build's own host-compiled restatement):
      * env step = the HOST build of the product's own single-source kernel text (oracle/host/hx_host.cpp: hx_math.h + hx_dyn.h +
        hx_env.h compiled with g++ -O3 -fopenmp, OpenMP over robots);
      * learner = the compiled host restatement of the learner's dense arithmetic (oracle/host/hx_learner_host.cpp: AVX2-FMA GEMM,
        ELU, Adam; OpenMP) under the numpy oracle's PPO (loss head, GAE, schedule), parity-checked against the reference's own
        PPO outputs (tests/test_host_learner.py, tests/golden/ppo_small.npz).
    Timed: `sample_steps` env steps of `sample_envs` robots on the same terrain (all host threads, then OMP_NUM_THREADS = 10, the
    reference's physx.num_threads, hector_config.py:109); 15 policy steps at 4096 rows + ONE full-size minibatch step
    (forward, loss, backward, clip, Adam on `update_rows` = 61 440 rows) scaled to the 8 minibatch steps of an iteration;
    combined into env-steps/s of a whole iteration.  For scale: the reference's own torch-CPU learner does 18.1 k env-steps/s
    on 8 cores; its env cannot run anywhere (PhysX)."""
    import subprocess
    env_code = (
        "import sys, time, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from isaac_amd.envs.configs import HectorCfg\n"
        "from oracle.host import HostEnv, lib\n"
        f"cfg = HectorCfg(); cfg.env.num_envs = {sample_envs}; cfg.terrain.mesh_type = {terrain!r}; cfg.seed = 5\n"
        "np.random.seed(5)\n"
        "env = HostEnv(cfg)\n"
        f"a = (0.3 * np.random.default_rng(0).standard_normal(({sample_envs}, 10))).astype(np.float32)\n"
        "for _ in range(10): env.L.hxh_step(env.h, a.ctypes.data, None)\n"
        "best = 1e30\n"
        "for rep in range(3):\n"
        "    t0 = time.perf_counter()\n"
        f"    for _ in range({max(1, sample_steps // 3)}): env.L.hxh_step(env.h, a.ctypes.data, None)\n"
        f"    best = min(best, (time.perf_counter() - t0) / {max(1, sample_steps // 3)})\n"
        "print(json.dumps(dict(s_per_step=best, threads=int(lib().hxh_num_threads()))))\n")
    N, T = 4096, update_rows // 4096
    learner_code = (
        "import sys, time, json, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from oracle.ppo import ActorCriticOracle\n"
        "from oracle.host.learner import HostPPOOracle, host_actor_critic, lib\n"
        f"N, T = {N}, {T}\n"
        "rng = np.random.default_rng(0)\n"
        "ac = host_actor_critic(ActorCriticOracle.default_init(rng))\n"
        "alg = HostPPOOracle(ac, N, T, num_learning_epochs=1, num_mini_batches=1)\n"
        "O, P, E = (rng.standard_normal((N, k)).astype(np.float32) for k in (615, 1050, 10))\n"
        "for rep in range(2):                              # first pass = warm-up: thread pool, first touch of storage and buffers\n"
        "    alg.step = 0\n"
        "    t0 = time.perf_counter()\n"
        "    for t in range(T):\n"
        "        alg.act(O, P, E); alg.process_env_step(np.full(N, 0.02, np.float32), np.zeros(N, bool))\n"
        "    t_act = (time.perf_counter() - t0) / (N * T)\n"
        "alg.compute_returns(P)\n"
        "perm = rng.permutation(N * T)\n"
        "alg.update(perm); alg.step = T                    # warm-up: first touch of the workspaces\n"
        "t0 = time.perf_counter(); alg.update(perm)\n"
        "t_upd = (time.perf_counter() - t0) / (N * T)\n"
        "print(json.dumps(dict(t_act=t_act, t_upd_row=t_upd, threads=int(lib().hxl_num_threads()))))\n")

    ncpu = host_cpus()

    def run(code, threads, **extra):
        env = dict(os.environ, OMP_WAIT_POLICY="passive", **extra)
        env["OMP_NUM_THREADS"] = str(threads if threads else ncpu)          # "all cores" = all this job may use (host_cpus)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
        if out.returncode != 0:
            raise RuntimeError(out.stderr[-600:])
        return json.loads(out.stdout.strip().splitlines()[-1])

    full, ten = run(env_code, None), run(env_code, min(10, ncpu))
    # numpy's BLAS pool is kept to one thread here: its spinning workers would fight the OpenMP team of the compiled learner
    lf, lt = run(learner_code, None, OPENBLAS_NUM_THREADS="1"), run(learner_code, min(10, ncpu), OPENBLAS_NUM_THREADS="1")
    n = sample_envs
    per_env_step = lambda env_s, l: env_s / n + l["t_act"] + 2 * l["t_upd_row"]          # 2 epochs over every stored row
    v_full, v_ten = 1.0 / per_env_step(full["s_per_step"], lf), 1.0 / per_env_step(ten["s_per_step"], lt)
    learner_only = lambda l: 1.0 / (l["t_act"] + 2 * l["t_upd_row"])
    return {"value": v_full, "unit": "env-steps/s", "cores": int(full["threads"]), "kind": "port",
            "sample": f"{sample_steps} env steps x {n} robots (terrain {terrain}) on the host build of the kernel source: "
                      f"{1e3 * full['s_per_step']:.1f} ms per step with {full['threads']} OpenMP threads, {1e3 * ten['s_per_step']:.1f} ms with 10; "
                      f"learner = compiled host restatement (AVX2 GEMM + numpy loss head), {T} policy steps at {N} rows + one minibatch step on {N * T} rows, "
                      f"scaled to 2 epochs: {1e6 * lf['t_act']:.2f} + 2 x {1e6 * lf['t_upd_row']:.2f} us per env-step with {lf['threads']} threads; "
                      f"this job may use {ncpu} of the host's {os.cpu_count()} logical cores (affinity / cgroup quota)",
            "env_only_env_steps_per_s": n / full["s_per_step"], "learner_only_env_steps_per_s": learner_only(lf),
            "omp10": {"value": v_ten, "env_only_env_steps_per_s": n / ten["s_per_step"], "learner_only_env_steps_per_s": learner_only(lt), "cores": min(10, ncpu)},
            "reference_learner_8_cores_env_steps_per_s": 18.1e3}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which itself never
    initialises HIP (no capi / build call has happened yet).  Rank 0's stdout -- the one JSON line -- is relayed to ours,
    every other stream goes to stderr.  A rank that dies takes the job down: the others are given 20 s to notice (their
    collectives time out by themselves, include/hx_ppo.h) and are then terminated.  Exit code = the first non-zero one."""
    import subprocess
    # Build once, here: hipcc / make only -- nothing in isaac_amd.build or oracle.host.build touches HIP -- so the ranks find
    # current binaries and skip their own build (ISAAC_BENCH_PREBUILT).
    from isaac_amd import build as hx_build
    hx_build.build()
    try:
        import shutil
        if shutil.which("g++") and shutil.which("make"):
            from oracle.host import build as build_host
            build_host()
    except Exception as e:
        print(f"warning: host build of the oracle failed ({e!r})", file=sys.stderr)
    # The rendezvous store lives in this process for the whole job (a launcher-style agent store, as under
    # torch.distributed.run): port 0 = the kernel picks a free port and this process keeps it, so no rank can lose a race for it.
    # Importing torch.distributed initialises no GPU.
    port, store = None, None
    if os.environ.get("HX_BENCH_CHILD_PROBE", "store") == "store":
        from datetime import timedelta
        from torch.distributed import TCPStore
        store = TCPStore("127.0.0.1", 0, n, True, timeout=timedelta(seconds=300), wait_for_workers=False)
        port = store.port
    else:                                   # the CPU probe test needs no store (and no torch import)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ISAAC_BENCH_PREBUILT="1")
        if store is not None:
            env["TORCHELASTIC_USE_AGENT_STORE"] = "True"     # every rank is a client of the store above (isaac_amd/parallel.py exchange_unique_id)
        else:
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad and deadline is None:
            rc, deadline = bad[0], time.time() + 20.0
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 1e9
        time.sleep(0.05)
    reader.join(timeout=10)
    for p in procs:
        if p.returncode != 0 and rc == 0:
            rc = p.returncode
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode() if isinstance(out0[0], bytes) else out0[0])
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--task", default="hector", choices=["hector", "hector_full", "humanoid_ppo"],
                    help="hector is BASELINE.json's metric config; hector_full (18 DoF) and humanoid_ppo (XBot-L, 12 DoF) are the sibling "
                         "tasks of SURVEY 8f-4, side measurements")
    ap.add_argument("--shards", type=int, default=1, help="env shards per GPU driven round-robin on separate streams (1 = off; measured slower than the deferred-critic overlap, see DESIGN.md)")
    ap.add_argument("--terrain", default="trimesh", choices=["trimesh", "heightfield", "plane"],
                    help="terrain.mesh_type; 'trimesh' is the reference's default for the hector task (hector_config.py:45)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="learner precision: f32 = the metric's configuration (BASELINE config 2); bf16 = BASELINE config 4 "
                         "(forward/dgrad products on the bf16 matrix cores, fp32 master weights and wgrads) -- a different "
                         "configuration, reported with dtype 'bf16' and never as the headline")
    ap.add_argument("--storage", default="auto", choices=["auto", "frames", "rows"],
                    help="rollout storage of the observations: frames = every robot's 41 / 70-wide frames once, rows = the reference's "
                         "stacked 615 / 1050-wide rows (stacking and gather launches), auto = by measurement (frames from 16 384 robots "
                         "per GPU up; isaac_amd/algo/on_policy_runner.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic, one rank only: walk the N > 1 code path (RCCL all-reduce per optimiser step, as identity) "
                         "to see what the distributed orchestration costs before any link time")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket GEMM launches with HIP events")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))          # launcher mode: nothing below runs in this process
    if os.environ.get("HX_BENCH_CHILD_PROBE"):            # tests/test_parallel_cpu.py: what a spawned rank sees, without a GPU
        seen = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ISAAC_BENCH_PREBUILT", "TORCHELASTIC_USE_AGENT_STORE")}
        if os.environ.get("HX_BENCH_CHILD_PROBE") == "store":          # the unique-id rendezvous against the launcher's store, as HxComm does it
            from isaac_amd.parallel import exchange_unique_id
            rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            seen["ID"] = exchange_unique_id(rank, world, lambda: bytes([7 + rank]) * 128, "probe_uid").hex()[:8]
        print(json.dumps(seen), flush=True)
        if os.environ.get("HX_BENCH_CHILD_PROBE") == "fail" + os.environ.get("RANK", ""):
            raise SystemExit(3)
        if os.environ.get("HX_BENCH_CHILD_PROBE", "").startswith("fail"):
            time.sleep(600)                               # a healthy rank waiting for a peer that died: the launcher must end it
        raise SystemExit(0)

    import contextlib
    # stdout carries exactly one line: the result JSON.  Python-level prints go to stderr, and so does file descriptor 1
    # itself while the job runs: RCCL prints a version banner to fd 1 from native code when its first communicator comes up.
    sys.stdout.flush()
    real_fd = os.dup(1)
    os.dup2(2, 1)
    real_stdout = os.fdopen(real_fd, "w")
    sys.stdout = sys.stderr
    if os.environ.get("ISAAC_BENCH_PREBUILT") != "1":        # a rank started by spawn_ranks: the parent has built everything
        import __graft_entry__
        __graft_entry__.build()
    from isaac_amd import capi
    from isaac_amd.parallel import init_comm
    from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO, HectorFullCfg, HectorFullCfgPPO, XBotLCfg, XBotLCfgPPO
    from isaac_amd.envs.hector_env import HectorFreeEnv, HectorFullFreeEnv, PipelinedHectorEnv, XBotLFreeEnv, class_to_dict
    from isaac_amd.algo.on_policy_runner import OnPolicyRunner
    from isaac_amd.utils.helpers import set_seed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and args.gpus != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus {args.gpus}` (self-spawning) or "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    if args.force_collectives and world == 1:
        from isaac_amd.parallel import HxComm
        comm = HxComm(rank=0, world_size=1, local_rank=0)      # one rank, real RCCL: every collective of the N > 1 path runs
        comm.force_collectives = True
    else:
        comm = init_comm()
    ndev = max(1, capi.lib().hx_device_count())
    local = comm.local_rank % ndev       # one rank per GPU under the driver; wraps only in the one-GPU gloo-staged rehearsal
    capi.check(capi.lib().hx_set_device(local), "hx_set_device")

    full = args.task != "hector"              # a sibling task: no cpu_baseline / traffic bookkeeping of the headline config
    if full and args.shards > 1:
        raise SystemExit("--shards applies to the hector task only")
    env_cfg, train_cfg = {"hector": (HectorCfg, HectorCfgPPO), "hector_full": (HectorFullCfg, HectorFullCfgPPO),
                          "humanoid_ppo": (XBotLCfg, XBotLCfgPPO)}[args.task]
    env_cfg, train_cfg = env_cfg(), train_cfg()
    env_cfg.env.num_envs = args.envs
    env_cfg.terrain.mesh_type = args.terrain
    env_cfg.seed = set_seed(train_cfg.seed + comm.rank)
    if args.shards > 1:
        env = PipelinedHectorEnv(env_cfg, sim_device=f"cuda:{local}", headless=True, num_shards=args.shards)
    else:
        env = {"hector": HectorFreeEnv, "hector_full": HectorFullFreeEnv, "humanoid_ppo": XBotLFreeEnv}[args.task](env_cfg, sim_device=f"cuda:{local}", headless=True)
    tcfg = class_to_dict(train_cfg)
    if args.dtype != "f32":
        tcfg["algorithm"]["mlp_dtype"] = args.dtype          # extra PPO keyword of this build (hx_ppo_set_compute_dtype)
    tcfg["runner"]["observation_storage"] = args.storage
    runner = OnPolicyRunner(env, tcfg, log_dir=None, device=f"cuda:{local}", comm=comm)
    T = runner.num_steps_per_env

    # HIP-event brackets cost GPU time (every GEMM launch bracketed: -1.5 % env-steps/s), so the full per-kernel table is
    # taken during the untimed warm-up iterations and only the dominant kernel found there is bracketed in the timed region.
    prof_all = None
    env_step_ms, env_step_from = None, None
    time_env = (not args.no_prof) and args.shards == 1          # the env-step kernel's brackets: same rule, warm-up when there is one
    if not args.no_prof and args.warmup > 0:
        runner.alg.prof_begin()
        if time_env:
            capi.check(capi.lib().hx_sim_time(env._h, 1, None), "hx_sim_time")
    runner.learn(args.warmup, init_at_random_ep_len=True)          # untimed warm-up iterations
    env.sync()
    if not args.no_prof and args.warmup > 0:
        prof_all = runner.alg.prof_end()
        if time_env:
            env_step_ms = np.zeros(2, np.float64)
            capi.check(capi.lib().hx_sim_time(env._h, 0, env_step_ms.ctypes.data), "hx_sim_time")
            env_step_from, time_env = "warm-up iterations", False
    comm.barrier()
    sample_every, per_iter = 1, 1
    if not args.no_prof:
        dom = max(prof_all["kernels"], key=lambda r: r["ms"]) if prof_all and prof_all["kernels"] else None
        dominant = dom["name"] if dom else None
        # A uniform sample of the dominant symbol's launches: an event pair idles the stream ~7 us on either side of the launch
        # (kernel trace: 0 us between unbracketed launches), 0.6 ms per iteration when 48 launches carry one.  The stride is
        # coprime to the symbol's launches per iteration, so the sample walks through every shape that shares the symbol
        # (the two weight-gradient groups of a minibatch alternate; a stride of 3 visits both).
        import math
        per_iter = max(1, dom["launches"] // max(1, args.warmup)) if dom else 1
        sample_every = next((s for s in (3, 5, 7, 11, 13) if math.gcd(s, per_iter) == 1), 1) if per_iter >= 6 else 1
        runner.alg.prof_begin(only=dominant, sample_every=sample_every)
        if time_env:
            capi.check(capi.lib().hx_sim_time(env._h, 1, None), "hx_sim_time")
    t0 = time.perf_counter()
    runner.learn(args.steps, init_at_random_ep_len=False)
    env.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    # per-iteration times as the runner's own clock saw them (collection + learning): the median beside the mean
    iter_s = sorted(getattr(runner, "iteration_times", [])[-args.steps:])
    prof = None if args.no_prof else runner.alg.prof_end()
    if time_env:
        env_step_ms = np.zeros(2, np.float64)
        capi.check(capi.lib().hx_sim_time(env._h, 0, env_step_ms.ctypes.data), "hx_sim_time")
        env_step_from = "timed region"
    elapsed = comm.max_over_ranks(elapsed)

    if comm.rank == 0:
        env_steps = T * args.envs * world * args.steps
        value = env_steps / elapsed
        dims = lambda d: "[" + ",".join(str(x) for x in d) + "]"
        out = {"metric": f"env-steps/sec (whole node), {args.task} {args.envs} envs/GPU", "value": value, "unit": "env-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
               "ms_per_step_median": (1e3 * iter_s[len(iter_s) // 2]) if iter_s else None,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{args.task} {args.envs} envs/GPU, 1 iteration = 60 env steps (10 x 1 ms substeps) + PPO "
                                      f"update {train_cfg.algorithm.num_learning_epochs} epochs x {train_cfg.algorithm.num_mini_batches} minibatches, fp32 HIP sim + MLP actor {dims(train_cfg.policy.actor_hidden_dims)} / "
                                      f"critic {dims(train_cfg.policy.critic_hidden_dims)}"
                                      + ("" if args.dtype == "f32" else " (bf16 forward/dgrad MFMA, fp32 master weights)"),
                          "num_envs_per_gpu": args.envs, "num_steps_per_env": T, "parallelism": f"dp{world}",
                          "terrain": args.terrain, "env_shards": args.shards, "observation_storage": ("frames" if getattr(runner.alg, "frames", None) else "rows"), **({"forced_collectives": True} if args.force_collectives else {}), "collection_s": runner.last_perf.get("collection_time"),
                          "learn_s": runner.last_perf.get("learn_time")}}
        if prof is not None and prof["kernels"]:
            k = max(prof["kernels"], key=lambda r: r["ms"])
            achieved = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            build_id = capi.lib().hx_build_id().decode()
            traffic, traffic_detail = pmc_traffic(k["name"], build_id) if (args.envs == 4096 and args.terrain == "trimesh" and args.shards == 1 and not full) else (None, None)
            # bf16 mode: the same kernel ids run on the bf16 matrix cores (dense peak 2.5 PFLOP/s); with fp32 operands in HBM
            # those products are memory-bound, which is what the small fraction of that peak says
            peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else 2500.0
            out["roofline"] = {"bound": "mfma", "kernel": k["name"] + ("" if args.dtype == "f32" else " [bf16 operands]"), "achieved": achieved, "peak": peak,
                               "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic if args.dtype == "f32" else None,
                               "traffic_detail": traffic_detail,
                               "build_id": build_id, "launches": k["launches"], "launches_bracketed": f"1 launch in {sample_every} of the symbol in the timed region ({per_iter} launches per iteration; the stride is coprime to that, so every shape sharing the symbol is sampled)" if sample_every > 1 else "all", "avg_launch_us": 1e3 * k["ms"] / max(1, k["launches"]),
                               "flop_per_launch": k["flops"] / max(1, k["launches"]),
                               "all_gemm_kernels": (prof_all or prof)["kernels"],
                               "all_gemm_kernels_from": "warm-up iterations (every launch on the learner's stream bracketed; the deferred critic's background launches overlap the rollout and are not)" if prof_all else "timed region",
                               "whole_iteration_mfma_frac": None if full else value / world * FLOP_PER_ENV_STEP / (peak * 1e12)}
        if env_step_ms is not None and env_step_ms[1] > 0 and "roofline" in out:
            # the env-step kernel: top kernel of the rollout, bound by VALU issue (it reads and writes ~1.3 KB per robot).
            # peak = one wave64 VALU instruction per SIMD every 4 cycles at 2.4 GHz on 1024 SIMDs
            us = 1e3 * env_step_ms[0] / env_step_ms[1]
            pmc = None if full else pmc_env_step(capi.lib().hx_build_id().decode())
            peak = 1024 * 2.4e9 / 4
            es = {"kernel": "hx_env_step_kernel", "bound": "valu-issue", "launches": int(env_step_ms[1]), "avg_launch_us": us,
                  "measured_in": env_step_from, "waves": (args.envs + 7) // 8, "peak_wave_insts_per_s": peak, "valu_insts_per_launch": None, "achieved_wave_insts_per_s": None, "frac": None}
            if pmc and args.envs == 4096 and args.terrain == pmc.get("terrain", "trimesh"):
                es["valu_insts_per_launch"] = pmc["valu_insts_per_launch"]
                es["achieved_wave_insts_per_s"] = pmc["valu_insts_per_launch"] / (us * 1e-6)
                es["frac"] = es["achieved_wave_insts_per_s"] / peak
                es["pmc_source"] = pmc.get("source")
            out["roofline"]["env_step"] = es
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
