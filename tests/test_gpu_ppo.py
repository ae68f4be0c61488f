"""GPU: the HIP learner against (a) the golden outputs of the reference's own PPO (tests/golden/ppo_*.npz)
and (b) the numpy oracle stage by stage, through the C ABI (isaac_amd.algo.ppo -> libhx.so).

Tolerances (fp32; MFMA f32 sums k in a different order than MKL/OpenBLAS):
  forward values / actions / log-probs  5e-5 absolute      returns, normalised advantages  1e-4
  losses 1e-4 relative, grad-norms 2e-4 relative, learning-rate schedule exact
  parameters after 8 Adam steps: |delta - delta_ref| sums within 2e-3 relative, 64-element slices 2e-6 absolute
"""
import os

import numpy as np
import pytest

from isaac_amd import capi
from isaac_amd.algo.ppo import PPO, ActorCritic
from oracle.ppo import ActorCriticOracle, PPOOracle
from tests.ppo_inputs import rollout_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _make(seed, T, N, lr, epochs=2, nmb=4):
    init = ActorCriticOracle.default_init(np.random.default_rng(seed))
    ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128], init_noise_std=1.0)
    ac.load_state_dict(init.state_dict())
    alg = PPO(ac, num_learning_epochs=epochs, num_mini_batches=nmb, clip_param=0.2, gamma=0.994, lam=0.9, value_loss_coef=1.0,
              entropy_coef=0.001, learning_rate=lr, max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive",
              desired_kl=0.01)
    alg.init_storage(N, T, [615], [1050], [10])
    return init, ac, alg


@pytest.mark.parametrize("name", ["ppo_small", "ppo_clip"])
def test_ppo_matches_reference_fixture(hxlib, name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    seed, T, N, epochs, nmb = (int(x) for x in fx["meta"])
    init, ac, alg = _make(seed, T, N, float(fx["lr0"]), epochs, nmb)
    inp = rollout_inputs(seed, T, N)
    # state_dict round trip through the padded device layout
    sd = ac.state_dict()
    for k, v in init.state_dict().items():
        np.testing.assert_array_equal(sd[k], v)
    for t in range(T):
        a = alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t]).numpy()
        np.testing.assert_allclose(a, fx["actions"][t], rtol=0, atol=5e-5)
        alg.process_env_step(inp["rewards"][t] * np.float32(fx["scale_rewards"]), inp["dones"][t].astype(np.uint8),
                             {"time_outs": inp["time_outs"][t].astype(np.uint8)})
    alg.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.buffer(1, (T, N)).numpy(), fx["values"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(alg.buffer(2, (T, N)).numpy(), fx["logp"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(alg.buffer(3, (T, N, 10)).numpy(), fx["mu"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(alg.buffer(4, (T, N)).numpy(), fx["stored_rewards"], rtol=0, atol=1e-5 * float(fx["scale_rewards"]))
    np.testing.assert_allclose(alg.buffer(5, (T, N)).numpy(), fx["returns"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(alg.buffer(6, (T, N)).numpy(), fx["advantages"], rtol=1e-4, atol=2e-4)
    mvl, msl = alg.update(perm=fx["perm"])
    assert abs(mvl - float(fx["mean_value_loss"])) <= 1e-4 * max(1.0, abs(float(fx["mean_value_loss"])))
    assert abs(msl - float(fx["mean_surrogate_loss"])) <= 1e-4
    assert abs(alg.learning_rate / float(fx["final_lr"]) - 1.0) < 1e-6
    sd = ac.state_dict()
    init_sd = init.state_dict()
    for k in sd:
        d = sd[k].astype(np.float64) - init_sd[k].astype(np.float64)
        ref_abs = float(fx["delta_abs_" + k])
        assert abs(np.abs(d).sum() - ref_abs) <= 2e-3 * ref_abs + 1e-9, k
        np.testing.assert_allclose(sd[k].reshape(-1)[:64], fx["slice_" + k], rtol=0, atol=4e-6, err_msg=k)
        np.testing.assert_allclose(sd[k].reshape(-1)[-64:], fx["slice_end_" + k], rtol=0, atol=4e-6, err_msg=k)
    m, v, step = alg.optimizer_state()
    assert step == int(fx["adam_step"])
    np.testing.assert_allclose(m[:10], fx["adam_m_std"], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(v[:10], fx["adam_v_std"], rtol=4e-3, atol=1e-10)


def test_ppo_stagewise_vs_oracle(hxlib):
    """One minibatch at a time against the numpy oracle: gradient buffer, KL, LR decision, grad norm."""
    seed, T, N = 21, 6, 32
    init, ac, alg = _make(seed, T, N, 1e-4, epochs=1, nmb=2)
    orc = PPOOracle(ActorCriticOracle.default_init(np.random.default_rng(seed)), N, T, num_learning_epochs=1,
                    num_mini_batches=2, learning_rate=1e-4)
    inp = rollout_inputs(seed, T, N)
    for t in range(T):
        alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t])
        orc.act(inp["obs"][t], inp["priv"][t], inp["eps"][t])
        alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        orc.process_env_step(inp["rewards"][t], inp["dones"][t], inp["time_outs"][t])
    alg.compute_returns(inp["priv"][T])
    orc.compute_returns(inp["priv"][T])
    np.testing.assert_allclose(alg.buffer(6, (T, N)).numpy(), orc.advantages, rtol=1e-4, atol=2e-4)
    perm = np.random.default_rng(3).permutation(T * N).astype(np.int32)
    L = hxlib
    dperm = capi.DeviceBuffer.from_host(perm)
    capi.check(L.hx_ppo_update_begin(alg._h, dperm.ptr), "begin")
    g, cnt = capi.C.c_void_p(), capi.C.c_int64()
    mbs = T * N // 2
    for i in range(2):
        capi.check(L.hx_ppo_minibatch_backward(alg._h, i, capi.C.byref(g), capi.C.byref(cnt)), "bwd")
        flat = capi.download(g.value, np.float32, (cnt.value,))
        info, grads = orc.loss_and_grads(perm[i * mbs:(i + 1) * mbs])
        # unpack the padded device layout with the same map the library uses (std is last on the device)
        stats = flat[-4:]
        assert abs(stats[0] / stats[3] - info["kl"]) < 2e-5 + 1e-3 * abs(info["kl"])
        assert abs(stats[1] / stats[3] - info["value"]) < 1e-4 * max(1, abs(info["value"]))
        assert abs(stats[2] / stats[3] - info["surrogate"]) < 1e-4
        gn_ref = np.sqrt(sum(float(np.sum(x.astype(np.float64) ** 2)) for x in grads))
        gn = np.sqrt(np.sum(flat[:-4].astype(np.float64) ** 2))
        assert abs(gn / gn_ref - 1) < 2e-4, (gn, gn_ref)
        orc.adapt_lr(info["kl"])
        orc.optimizer_step(grads)
        capi.check(L.hx_ppo_minibatch_step(alg._h, 1.0), "step")
        assert abs(alg.learning_rate / orc.lr - 1) < 1e-6
    sd = ac.state_dict()
    # Adam divides by sqrt(v): for weights whose gradient is at round-off level the normalised step is
    # noise, bounded by lr per step.  Require 99.9% of the elements within 3e-6 and all within 2 steps * lr.
    for k, v in orc.ac.state_dict().items():
        d = np.abs(sd[k] - v)
        assert d.max() <= 2 * 2.0 * orc.lr, (k, d.max())
        assert np.mean(d > 3e-6) < 1e-3, (k, float(np.mean(d > 3e-6)))


def test_device_permutation_is_a_permutation(hxlib):
    seed, T, N = 5, 4, 64
    _, _, alg = _make(seed, T, N, 1e-5)
    capi.check(hxlib.hx_ppo_update_begin(alg._h, None), "begin")
    p = alg.buffer(8, (T * N,), np.int32).numpy()
    assert sorted(p.tolist()) == list(range(T * N))
    assert not np.array_equal(p, np.arange(T * N))


def test_rollout_overflow_raises(hxlib):
    _, _, alg = _make(1, 2, 16, 1e-5)
    z = np.zeros((16, 615), np.float32), np.zeros((16, 1050), np.float32)
    for _ in range(2):
        alg.act(*z)
        alg.process_env_step(np.zeros(16, np.float32), np.zeros(16, np.uint8), {})
    with pytest.raises(RuntimeError, match="Rollout buffer overflow"):     # rollout_storage.py:88-89
        alg.act(*z)


def test_collective_code_path_single_rank(hxlib):
    """Walk the multi-GPU path with one rank: RCCL communicator inside libhx.so (hx_comm_init), parameter broadcast,
    ncclAllReduce of the gradient buffer and of the advantage moments on the learner's stream, 1/world scaling in the Adam
    kernel.  With world_size 1 the result must be bit-identical to the plain single-process path."""
    from isaac_amd.parallel import HxComm
    comm = HxComm(rank=0, world_size=1, local_rank=0)
    comm.force_collectives = True
    assert comm.max_over_ranks(3.25) == 3.25 and comm.sum_over_ranks(-1.5) == -1.5
    seed, T, N = 31, 4, 64
    inp = rollout_inputs(seed, T, N)
    perm = np.random.default_rng(1).permutation(T * N).astype(np.int32)
    results = []
    for c in (None, comm):
        init = ActorCriticOracle.default_init(np.random.default_rng(seed))
        ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
        ac.load_state_dict(init.state_dict())
        alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, gamma=0.994, lam=0.9, entropy_coef=0.001, learning_rate=1e-4,
                  schedule="adaptive", desired_kl=0.01, comm=c)
        alg.init_storage(N, T, [615], [1050], [10])
        for t in range(T):
            alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t])
            alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        alg.compute_returns(inp["priv"][T])
        losses = alg.update(perm=perm)
        results.append((losses, alg.learning_rate, ac.state_dict()))
        alg.close()
    (l0, lr0, sd0), (l1, lr1, sd1) = results
    assert l0 == l1 and lr0 == lr1
    for k in sd0:
        np.testing.assert_array_equal(sd0[k], sd1[k], err_msg=k)
    comm.close()


def test_ranks_draw_different_exploration_noise(hxlib):
    """hx_ppo_set_seed(seed + rank, seed): two learners standing for two data-parallel ranks, identical weights and identical
    observations, must sample DIFFERENT actions (independent exploration), and a learner re-created with the same rank's seed
    the same ones; hx_ppo_set_rng_state restores a position in the stream (what a resumed checkpoint does)."""
    class FakeComm:
        in_library, world_size, local_rank = False, 1, 0

        def __init__(self, rank):
            self.rank = rank
    rng = np.random.default_rng(0)
    o, p = rng.standard_normal((32, 615)).astype(np.float32), rng.standard_normal((32, 1050)).astype(np.float32)
    init = ActorCriticOracle.default_init(np.random.default_rng(5))
    acts = []
    for rank in (0, 1, 0):
        ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
        ac.load_state_dict(init.state_dict())
        alg = PPO(ac, comm=FakeComm(rank), seed=5)
        alg.init_storage(32, 4, [615], [1050], [10])
        a0 = alg.act(o, p).numpy().copy()
        alg.process_env_step(np.zeros(32, np.float32), np.zeros(32, np.uint8), {})
        a1 = alg.act(o, p).numpy().copy()
        if rank == 1:
            alg.process_env_step(np.zeros(32, np.float32), np.zeros(32, np.uint8), {})
            alg.load_rng_state(0, 0)                     # rewind: the first draw comes back
            np.testing.assert_array_equal(alg.act(o, p).numpy(), a0)
        acts.append((a0, a1))
        alg.close()
    assert np.abs(acts[0][0] - acts[1][0]).max() > 0.1                   # rank 0 vs rank 1: different noise
    assert np.abs(acts[0][0] - acts[0][1]).max() > 0.1                   # successive calls: different noise
    np.testing.assert_array_equal(acts[0][0], acts[2][0])                # same rank, same seed: same stream
    np.testing.assert_array_equal(acts[0][1], acts[2][1])


def test_non_finite_gradient_skips_the_step(hxlib):
    """A NaN that reaches the loss (here: a NaN reward) must not reach the weights: the optimiser step is skipped."""
    seed, T, N = 4, 4, 32
    init, ac, alg = _make(seed, T, N, 1e-3, epochs=1, nmb=2)
    rng = np.random.default_rng(0)
    for t in range(T):
        o, p = rng.standard_normal((N, 615)).astype(np.float32), rng.standard_normal((N, 1050)).astype(np.float32)
        alg.act(o, p, eps=rng.standard_normal((N, 10)).astype(np.float32))
        r = rng.uniform(0, 0.05, N).astype(np.float32)
        if t == 2:
            r[5] = np.nan
        alg.process_env_step(r, np.zeros(N, np.uint8), {})
    alg.compute_returns(p)
    before = {k: v.copy() for k, v in ac.state_dict().items()}
    alg.update()
    after = ac.state_dict()
    for k in before:
        assert np.array_equal(before[k], after[k]) and np.isfinite(after[k]).all(), k
    alg.close()


def test_grouped_launches_change_scheduling_only(hxlib, monkeypatch):
    """The update's grouped launches (hx_gemm_group_kernel) at a size where they are active (N = 1024, T = 64: 4 minibatches
    of 16 384 rows, 2 epochs).  HX_GEMM_PAIR only moves layer l of the actor and of the critic into one launch -- the same
    tiles with the same k order -- so parameters and losses must be BIT-identical with and without it.  HX_WGRAD_GROUP cuts
    the 16 384-row reduction of the weight gradients into other slices (one slice count per group instead of one per layer):
    same sums in another association, so the eight Adam steps agree to fp32 round-off, not to the bit; HX_HEAD_MFMA likewise."""
    seed, T, N = 41, 64, 1024
    inp = rollout_inputs(seed, T, N)
    perm = np.random.default_rng(1).permutation(T * N).astype(np.int32)

    def run(env):
        for k in ("HX_GEMM_PAIR", "HX_WGRAD_GROUP", "HX_WGRAD_MULTI", "HX_HEAD_MFMA"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        _, ac, alg = _make(seed, T, N, 1e-4)
        for t in range(T):
            alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t])
            alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        alg.compute_returns(inp["priv"][T])
        losses = alg.update(perm=perm)
        out = (losses, alg.learning_rate, ac.state_dict())
        alg.close()
        return out

    base = run({})
    unpaired = run({"HX_GEMM_PAIR": "0"})
    assert base[0] == unpaired[0] and base[1] == unpaired[1]
    for k in base[2]:
        np.testing.assert_array_equal(base[2][k], unpaired[2][k], err_msg=k)
    # the same for the loss head on the matrix cores (hx_loss_head_mfma_kernel) against the VALU kernel it replaces for
    # hector-shaped heads: the dot products, the sum over actions and the head's weight gradients are associated differently
    # ... and for the weight gradients: one workgroup per CU with per-product tile shapes (default), the grouped split-K launches of
    # round 3 (HX_WGRAD_MULTI=0), one launch per layer (HX_WGRAD_MULTI=0 HX_WGRAD_GROUP=0): three ways to cut the same reductions
    for other in ({"HX_WGRAD_MULTI": "0"}, {"HX_WGRAD_MULTI": "0", "HX_WGRAD_GROUP": "0"}, {"HX_HEAD_MFMA": "0"}):
        alt = run(other)
        assert abs(base[0][0] - alt[0][0]) < 1e-5 * max(1.0, abs(base[0][0])) and abs(base[0][1] - alt[0][1]) < 1e-5, other
        assert base[1] == alt[1], other
        for k in base[2]:
            d = np.abs(base[2][k] - alt[2][k])
            assert d.max() < 5e-6 and np.mean(d > 2e-7) < 1e-2, (other, k, float(d.max()), float(np.mean(d > 2e-7)))


def test_launch_profiler_rows_and_sampling(hxlib):
    """include/hx_lab.h: one row per kernel symbol, named as rocprofv3 prints it; selecting a symbol keeps only its launches;
    sample_every = n brackets every n-th of them (what bench.py does in its timed region), and the flops of a row are those
    of the bracketed launches -- so TFLOP/s of a sample is comparable with that of the full set."""
    seed, T, N = 43, 16, 256
    inp = rollout_inputs(seed, T, N)
    perm = np.random.default_rng(1).permutation(T * N).astype(np.int32)
    _, ac, alg = _make(seed, T, N, 1e-4)

    def iteration():
        for t in range(T):
            alg.act(inp["obs"][t], inp["priv"][t], eps=inp["eps"][t])
            alg.process_env_step(inp["rewards"][t], inp["dones"][t].astype(np.uint8), {"time_outs": inp["time_outs"][t].astype(np.uint8)})
        alg.compute_returns(inp["priv"][T])
        alg.update(perm=perm)

    alg.prof_begin()
    iteration()
    full = {k["name"]: k for k in alg.prof_end()["kernels"]}
    grouped = [n for n in full if n == "hx_wgrad_multi_kernel"]
    assert len(grouped) == 1, sorted(full)                      # the weight gradients: 2 epochs x 4 minibatches, >= 1 launch each
    sym = grouped[0]
    total = full[sym]["launches"]
    assert total >= 8 and full[sym]["flops"] > 0 and full[sym]["ms"] > 0
    alg.prof_begin(only=sym, sample_every=3)
    iteration()
    part = alg.prof_end()["kernels"]
    assert [k["name"] for k in part] == [sym]
    assert part[0]["launches"] == (total + 2) // 3                # launches 0, 3, 6, ... of the symbol
    per_launch = full[sym]["flops"] / total
    assert abs(part[0]["flops"] / part[0]["launches"] / per_launch - 1.0) < 0.35      # a sample of the same mix of launch shapes
    alg.close()
