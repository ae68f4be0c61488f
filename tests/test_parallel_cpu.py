"""CPU, world_size 2 over gloo: the data-parallel orchestration (isaac_amd/parallel.py) keeps ranks
bit-identical and equals a single process training on the union of the shards."""
import os
import subprocess
import sys

import numpy as np

from isaac_amd.parallel import Comm, shard_envs
from oracle.ppo import ActorCriticOracle, PPOOracle
from tests.ppo_inputs import rollout_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_envs_partitions():
    for total, world in ((4096, 8), (10, 3), (7, 7)):
        spans = [shard_envs(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_identity_comm():
    c = Comm()
    assert c.world_size == 1 and c.max_over_ranks(3.5) == 3.5 and c.alloc_grad_buffer(10) is None


def test_two_ranks_equal_single_process(tmp_path):
    port = 29600 + os.getpid() % 300
    out = str(tmp_path / "rank{rank}.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = np.load(out.format(rank=0)), np.load(out.format(rank=1))
    # ranks end bit-identical: same broadcast init, same global statistics, same reduced gradient
    np.testing.assert_array_equal(r0["params"], r1["params"])
    np.testing.assert_array_equal(r0["hist"], r1["hist"])
    # single process over all 32 envs with the minibatches being the unions of the ranks' minibatches
    total, T, seed = 32, 4, 3
    ac = ActorCriticOracle.default_init(np.random.default_rng(100), 24, 40, 4, (32, 16, 8), (32, 16, 8))
    alg = PPOOracle(ac, total, T, num_learning_epochs=1, num_mini_batches=2, learning_rate=1e-3)
    inp = rollout_inputs(seed, T, total, 24, 40, 4)
    for t in range(T):
        alg.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        alg.process_env_step(inp["rewards"][t], inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.advantages[:, :16], r0["adv"], rtol=0, atol=2e-6)       # global normalisation
    np.testing.assert_allclose(alg.advantages[:, 16:], r1["adv"], rtol=0, atol=2e-6)
    n = 16
    perm = np.random.default_rng(9).permutation(n * T)
    mbs = n * T // 2
    for i in range(2):
        loc = perm[i * mbs:(i + 1) * mbs]                          # local flat index t*n + e  ->  global t*32 + e (+16)
        t, e = loc // n, loc % n
        idx = np.concatenate([t * total + e, t * total + e + n])
        info, grads = alg.loss_and_grads(idx)
        alg.adapt_lr(info["kl"])
        alg.optimizer_step(grads)
        assert abs(info["kl"] - r0["hist"][i][0]) < 1e-6 + 1e-4 * abs(info["kl"])
        assert abs(alg.lr - r0["hist"][i][1]) < 1e-12
    single = np.concatenate([p.reshape(-1) for p in ac.params()])
    d = np.abs(single - r0["params"])
    assert np.mean(d > 2e-6) < 5e-3 and d.max() < 4e-3, (d.max(), float(np.mean(d > 2e-6)))
