"""GPU: the N > 1 path with the real kernels.  Two ranks (sharing the one GPU of the test box, collectives staged
through gloo) train for two iterations on DIFFERENT robots; replicas must stay bit-identical -- parameters, Adam
moments, learning rate -- because every rank applies the same all-reduced gradient and statistics."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_stay_bit_identical(hxlib, tmp_path):
    port = 29700 + os.getpid() % 200
    out = str(tmp_path / "rank{rank}.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HX_DIST_BACKEND="gloo-staged", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out, "2"], env=env))
    for p in procs:
        assert p.wait(timeout=500) == 0
    r0, r1 = np.load(out.format(rank=0)), np.load(out.format(rank=1))
    assert not np.array_equal(r0["first_obs"], r1["first_obs"])          # the ranks really simulated different robots
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_array_equal(r0["m"], r1["m"])
    np.testing.assert_array_equal(r0["v"], r1["v"])
    assert int(r0["step"]) == int(r1["step"]) == 16 and float(r0["lr"]) == float(r1["lr"])
    assert np.all(np.isfinite(r0["params"]))


def test_two_ranks_equal_single_process_on_the_union(hxlib, tmp_path):
    """Two ranks on the env_range halves of ONE logical 512-robot batch == a single process on all 512 (tests/dp_gpu_union_worker.py):
    same robots, same action noise, same minibatches, global advantage moments, summed gradients -- within the tolerance of
    fp32 sums taken in a different order."""
    port = 29400 + os.getpid() % 200
    out = str(tmp_path / "r{rank}.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HX_DIST_BACKEND="gloo-staged", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_union_worker.py"), "rank", out], env=env))
    for p in procs:
        assert p.wait(timeout=500) == 0
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HX_DIST_BACKEND")}
    uout = str(tmp_path / "union{rank}.npz")
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_union_worker.py"), "union", uout], env=env, timeout=500).returncode == 0
    r0, r1, u = np.load(out.format(rank=0)), np.load(out.format(rank=1)), np.load(uout.format(rank=0))
    np.testing.assert_array_equal(r0["params"], r1["params"])
    # the second rollout ran with parameters that already differ by round-off, so only the first one's samples are bit-equal;
    # the last iteration's actions / advantages agree to the tolerance of that round-off
    np.testing.assert_allclose(np.concatenate([r0["actions"], r1["actions"]], axis=1), u["actions"], atol=2e-3)
    np.testing.assert_allclose(np.concatenate([r0["adv"], r1["adv"]], axis=1), u["adv"], atol=5e-3)
    np.testing.assert_allclose(r0["hist"], u["hist"], rtol=2e-3, atol=1e-6)          # losses and the learning-rate decisions
    d = np.abs(r0["params"] - u["params"])
    print("union vs 2 ranks: max |dparam| %.3g, share above 2e-6: %.4f" % (d.max(), float(np.mean(d > 2e-6))))
    assert d.max() < 2e-4 and np.mean(d > 2e-6) < 0.02, (d.max(), float(np.mean(d > 2e-6)))


def test_bench_self_spawns_two_ranks(hxlib):
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset): the parent starts both ranks before touching the GPU
    and relays rank 0's one JSON line.  The ranks share the test box's one GPU, so the transport is the gloo-staged rehearsal."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HX_DIST_BACKEND="gloo-staged", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=500, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["parallelism"] == "dp2"


def test_one_rank_rccl_path_equals_single_process(hxlib, tmp_path):
    """The measured N > 1 transport is RCCL inside libhx.so (hx_comm_init / hx_ppo_set_comm), which cannot run with two ranks
    on the one GPU of the test box.  With ONE rank it can: the whole distributed path (unique id, communicator, parameter
    broadcast, ncclAllReduce of the flat gradient + statistics buffer on the learner's stream in every optimiser step with
    1/world scaling, advantage-moment all-reduce) must then train bitwise like the single-process path."""
    outs = []
    for tag, extra in (("plain", {}), ("rccl", {"HX_DP_FORCE_RCCL": "1", "MASTER_ADDR": "127.0.0.1",
                                                "MASTER_PORT": str(29900 + os.getpid() % 90)})):
        out = str(tmp_path / (tag + "{rank}.npz"))
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        env.pop("HX_DIST_BACKEND", None)
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), out, "2"], env=env)
        assert p.wait(timeout=500) == 0, tag
        outs.append(np.load(out.format(rank=0)))
    a, b = outs
    np.testing.assert_array_equal(a["params"], b["params"])
    np.testing.assert_array_equal(a["m"], b["m"])
    np.testing.assert_array_equal(a["v"], b["v"])
    assert int(a["step"]) == int(b["step"]) == 16 and float(a["lr"]) == float(b["lr"])


def test_bench_prints_exactly_one_json_line(hxlib):
    """The driver's contract: stdout of bench.py is ONE JSON line.  Run it with the RCCL path on (one rank) -- RCCL prints a
    version banner to file descriptor 1 from native code, which must not reach stdout either."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + os.getpid() % 90))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HX_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "256", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--force-collectives"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=400, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert "destroy_process_group() was not called" not in r.stderr
