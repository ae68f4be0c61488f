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
