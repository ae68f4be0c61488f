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
