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


_RDV_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
from isaac_amd.parallel import exchange_unique_id
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
made = []
def make_id():
    made.append(1)
    return bytes(range(128))
raw = exchange_unique_id(rank, world, make_id, "hx_rccl_unique_id_0")
assert raw == bytes(range(128)), raw
assert len(made) == (1 if rank == 0 else 0)          # only rank 0 draws the id
raw2 = exchange_unique_id(rank, world, lambda: bytes(reversed(range(128))), "hx_rccl_unique_id_1")   # a second communicator of the job
assert raw2 == bytes(reversed(range(128)))
# the serving rank must outlive the readers (in the product ncclCommInitRank is that barrier)
from isaac_amd.parallel import _stores
st = _stores[-1][2]
if rank == 0:
    st.set("done0", b"1"); st.get("done1")
else:
    st.get("done0"); st.set("done1", b"1")
print("rank", rank, "ok", flush=True)
"""


def _run_rendezvous(agent_store):
    import socket
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    server = None
    if agent_store:                       # what torch.distributed.run's agent does: it serves the store, every rank is a client
        from datetime import timedelta
        from torch.distributed import TCPStore
        server = TCPStore("127.0.0.1", port, 2, True, timeout=timedelta(seconds=60), wait_for_workers=False)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        if agent_store:
            env["TORCHELASTIC_USE_AGENT_STORE"] = "True"
        procs.append(subprocess.Popen([sys.executable, "-c", _RDV_WORKER.format(root=ROOT)], env=env, stdout=subprocess.PIPE, text=True))
        if rank == 0:
            time.sleep(0.2)
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    del server


def test_unique_id_rendezvous_rank0_serves_the_store():
    _run_rendezvous(agent_store=False)


def test_unique_id_rendezvous_through_the_launcher_agent_store():
    _run_rendezvous(agent_store=True)


def test_hxcomm_refuses_a_rank_without_its_own_gpu():
    """RCCL needs one GPU per rank; LOCAL_RANK >= visible devices must be an error, not a silent modulo (and without any
    device the transport has no CPU form at all)."""
    import pytest
    from isaac_amd import capi
    from isaac_amd.parallel import HxComm
    ndev = capi.lib().hx_device_count()
    with pytest.raises(RuntimeError):
        HxComm(rank=0, world_size=1, local_rank=max(ndev, 1))


def test_bench_self_spawn_relays_rank0_and_propagates_failure():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (before any GPU / library call) and relays
    rank 0's one line; a rank that dies ends the job with its exit code and the surviving ranks are terminated."""
    import json
    import time
    env = dict(os.environ, HX_BENCH_CHILD_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and int(d["MASTER_PORT"]) > 0
    assert r.stderr.count('"RANK"') == 2          # the other ranks' stdout goes to stderr
    env["HX_BENCH_CHILD_PROBE"] = "fail1"         # rank 1 exits 3, ranks 0 and 2 would wait ten minutes for it
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 3 and time.time() - t0 < 60


def test_bench_launcher_serves_the_rendezvous_store_for_eight_ranks():
    """The self-spawning parent builds once (children see ISAAC_BENCH_PREBUILT), keeps the rendezvous port for the whole job (it
    serves the TCPStore itself, so no rank can lose a bind race) and every one of 8 children reads rank 0's unique id from it."""
    import json
    env = dict(os.environ, HX_BENCH_CHILD_PROBE="store")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(l) for l in (r.stdout + r.stderr).splitlines() if l.startswith("{") and '"RANK"' in l]
    assert sorted(int(d["RANK"]) for d in rows) == list(range(8))
    assert all(d["ISAAC_BENCH_PREBUILT"] == "1" and d["TORCHELASTIC_USE_AGENT_STORE"] == "True" for d in rows)
    assert len({d["MASTER_PORT"] for d in rows}) == 1
    assert all(d["ID"] == "07070707" for d in rows)          # rank 0's bytes reached every rank
