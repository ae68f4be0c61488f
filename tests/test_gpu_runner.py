"""GPU: the host surface end to end -- registry -> env -> runner.learn -> JSONL scalars -> checkpoint round trip."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_train_two_iterations_and_checkpoint_roundtrip(hxlib, tmp_path):
    from isaac_amd.envs import task_registry  # registers `hector`
    from isaac_amd.utils import get_args
    args = get_args(["--task=hector", "--headless", "--num_envs", "128", "--max_iterations", "2", "--seed", "3"])
    env, env_cfg = task_registry.make_env("hector", args=args)
    runner, train_cfg = task_registry.make_alg_runner(env, name="hector", args=args, log_root=str(tmp_path))
    runner.learn(2, init_at_random_ep_len=True)
    files = os.listdir(runner.log_dir)
    assert "model_0.pt" in files and "model_2.pt" in files and "scalars.jsonl" in files
    rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    assert len(rows) == 2
    for key in ("Loss/value_function", "Loss/surrogate", "Loss/learning_rate", "Policy/mean_noise_std", "Perf/total_fps",
                "Perf/collection time", "Perf/learning_time"):            # on_policy_runner.py:196-217
        assert key in rows[-1], key
    assert rows[-1]["Perf/total_fps"] > 0 and np.isfinite(rows[-1]["Loss/value_function"])
    # checkpoint format of on_policy_runner.py:278-287 and load :289-295
    import torch
    ck = torch.load(os.path.join(runner.log_dir, "model_2.pt"), map_location="cpu", weights_only=False)
    # the reference's four keys, plus the positions of the learner's counter-based random streams
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "iter", "infos", "hx_rng_state"} and ck["iter"] == 2
    assert ck["hx_rng_state"][0] >= 120 and ck["hx_rng_state"][1] == 2        # 2 x 60 act calls, 2 permutations
    assert list(ck["model_state_dict"])[0] == "std" and ck["model_state_dict"]["actor.0.weight"].shape == (512, 615)
    before = runner.alg.actor_critic.state_dict()
    m0, v0, step0 = runner.alg.optimizer_state()
    env2, _ = task_registry.make_env("hector", args=args)
    runner2, _ = task_registry.make_alg_runner(env2, name="hector", args=args, log_root=None)
    runner2.load(os.path.join(runner.log_dir, "model_2.pt"))
    after = runner2.alg.actor_critic.state_dict()
    for k in before:
        np.testing.assert_array_equal(before[k], after[k])
    m1, v1, step1 = runner2.alg.optimizer_state()
    assert step1 == step0 == 16
    np.testing.assert_array_equal(m0, m1)
    # inference policy (play.py:133-143 style)
    policy = runner2.get_inference_policy()
    act = policy(env2.get_observations()).numpy()
    assert act.shape == (128, 10) and np.all(np.isfinite(act))
    # the reference nn.Module layout accepts our state dict
    from isaac_amd.utils import export_policy_as_jit
    path = export_policy_as_jit(runner2.alg.actor_critic, str(tmp_path / "exported"))
    jit = torch.jit.load(path)
    obs = env2.get_observations().numpy()
    np.testing.assert_allclose(jit(torch.from_numpy(obs)).detach().numpy(), act, rtol=0, atol=2e-5)
    env.close(); env2.close()


def test_learning_signal_sanity(hxlib):
    """A few iterations on 256 robots: losses finite, episode bookkeeping alive, policy std stays positive."""
    from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO
    from isaac_amd.envs.hector_env import HectorFreeEnv, class_to_dict
    from isaac_amd.algo.on_policy_runner import OnPolicyRunner
    cfg = HectorCfg(); cfg.env.num_envs = 256; cfg.seed = 1
    env = HectorFreeEnv(cfg)
    runner = OnPolicyRunner(env, class_to_dict(HectorCfgPPO()), log_dir=None)
    runner.learn(3, init_at_random_ep_len=True)
    info, n_ep = env.episode_stats()
    assert n_ep > 0 and all(np.isfinite(v) for v in info.values())
    assert 0 < env.last_episode_length <= 2401
    assert np.all(runner.alg.actor_critic.std > 0)
    env.close()


def test_pipelined_runner_trains(hxlib):
    """The shard-pipelined rollout loop (bench.py --shards 2) produces a valid PPO iteration."""
    from isaac_amd.envs.configs import HectorCfg, HectorCfgPPO
    from isaac_amd.envs.hector_env import PipelinedHectorEnv, class_to_dict
    from isaac_amd.algo.on_policy_runner import OnPolicyRunner
    cfg = HectorCfg(); cfg.env.num_envs = 256; cfg.seed = 1
    env = PipelinedHectorEnv(cfg, num_shards=2)
    runner = OnPolicyRunner(env, class_to_dict(HectorCfgPPO()), log_dir=None)
    runner.learn(2, init_at_random_ep_len=True)
    info, n_ep = env.episode_stats()
    assert n_ep > 0 and all(np.isfinite(v) for v in info.values())
    sd = runner.alg.actor_critic.state_dict()
    assert all(np.all(np.isfinite(v)) for v in sd.values())
    assert runner.alg.buffer(1, (60, 256)).numpy().std() > 0          # both shards wrote their value columns
    v = runner.alg.buffer(1, (60, 256)).numpy()
    assert np.abs(v[:, :128]).sum() > 0 and np.abs(v[:, 128:]).sum() > 0
    env.close()


@pytest.mark.parametrize("task", ["hector", "humanoid_ppo"])
def test_frame_storage_equals_row_storage(hxlib, task):
    """Single-frame observation storage (include/hx_ppo.h hx_ppo_cfg.obs_frame; include/hx_sim.h hx_sim_step_frames) against the
    reference's stacked rows: the frame rings hold every robot's frames once, the fused actor / deferred critic / first-layer
    forward and weight-gradient products read their rows through (row start, first valid element) tables, there is no stacking
    launch and no minibatch gather -- the same numbers reach the same fma chains in the same order, so every stored array
    (rows expanded on request, actions, values, rewards, dones, the stale extras["time_outs"]), the env's own row buffers
    after the rollout AND the parameters after two updates must be equal bit for bit; the episode statistics up to the order
    of their atomic sums.  Robots start near their time limit so that resets (zeroed history) and the time-out bookkeeping occur
    inside the 16 steps.  humanoid_ppo: 47 x 15 / 73 x 3 frames, generic (unfused) rollout actor."""
    from isaac_amd import capi
    from isaac_amd.envs.configs import HectorCfg, XBotLCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv, XBotLFreeEnv
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.utils.helpers import set_seed
    import torch
    N, T = 200, 16                       # not a multiple of the actor's 16-row workgroups nor of the 8 robots per wave
    cfg_cls, env_cls = (HectorCfg, HectorFreeEnv) if task == "hector" else (XBotLCfg, XBotLFreeEnv)
    res = []
    for frames in (False, True):
        cfg = cfg_cls(); cfg.env.num_envs = N; cfg.seed = set_seed(9)
        if task != "hector":
            cfg.terrain.mesh_type = "plane"
        env = env_cls(cfg)
        ep = np.random.default_rng(1).integers(0, 2000, N).astype(np.int32)
        ep[::7] = int(env.max_episode_length) - 6 + (np.arange(len(ep[::7])) % 5)
        env.episode_length_buf = ep
        torch.manual_seed(3)
        # hector: the fused rollout actor (512 / 256 / 128); humanoid_ppo: other widths -> the layer-by-layer rollout actor
        dims = ([512, 256, 128], [768, 256, 128]) if task == "hector" else ([256, 128, 64], [256, 128, 64])
        ac = ActorCritic(env.num_obs, env.num_privileged_obs, env.num_actions, *dims)
        alg = PPO(ac, num_learning_epochs=2, num_mini_batches=4, gamma=0.994, lam=0.9, learning_rate=1e-5, schedule="adaptive",
                  desired_kl=0.01, stream=env.stream)
        alg.init_storage(N, T, [env.num_obs], [env.num_privileged_obs], [env.num_actions], obs_ld=env.obs_ld, priv_ld=env.priv_ld,
                         frames=env.frame_dims if frames else None)
        assert (alg.frames is not None) == frames
        out = {}
        for it in range(2):              # two rollouts: the second starts from the rows the first one left in the simulator
            alg.rollout([env], T)
            alg.compute_returns(env.get_privileged_observations())
            out.update({f"{k}{it}": alg.buffer(i, shp, dt).numpy().copy() for k, i, shp, dt in
                        (("actions", 0, (T, N, env.num_actions), np.float32), ("values", 1, (T, N), np.float32), ("logp", 2, (T, N), np.float32),
                         ("rewards", 4, (T, N), np.float32), ("returns", 5, (T, N), np.float32))})
            out[f"obs{it}"] = alg.storage_rows(capi.PPO_BUF_OBS).numpy().copy()
            out[f"priv{it}"] = alg.storage_rows(capi.PPO_BUF_PRIV).numpy().copy()
            out[f"dones{it}"] = alg.buffer(capi.PPO_BUF_DONES, (T, N), np.uint8).numpy().copy()
            out[f"timeouts{it}"] = alg.buffer(capi.PPO_BUF_TIMEOUTS, (T, N), np.uint8).numpy().copy()
            out[f"final_obs{it}"] = env.get_observations().numpy().copy()
            out[f"final_priv{it}"] = env.get_privileged_observations().numpy().copy()
            out[f"losses{it}"] = np.array(alg.update())
            out[f"params{it}"] = np.concatenate([v.reshape(-1) for v in ac.state_dict().values()])
        if frames:                       # the rows after the last step exist too (what the bootstrap value is computed from)
            last = alg.storage_rows(capi.PPO_BUF_PRIV, T, T + 1).numpy()[0]
            np.testing.assert_array_equal(last[:, :env.num_privileged_obs], out["final_priv1"])
        # a row-API step after a frame-mode rollout continues from the imported rows
        o2, p2, _, _, _ = env.step(np.zeros((N, env.num_actions), np.float32))
        out["next_obs"] = o2.numpy().copy()
        info, n_ep = env.episode_stats()
        out["stats"] = np.array([info[k] for k in sorted(info)] + [n_ep], np.float64)
        res.append(out)
        alg.close()
        env.close()
    a, b = res
    assert a["dones0"].sum() > 0 and a["timeouts0"].sum() > 0
    assert (a["obs1"][:, :, :env.obs_frame] == 0).all(axis=2).any(), "no zeroed history in the rollout: the test lost its point"
    for k in a:
        if k == "stats":      # sums of float atomics over the robots that reset: equal up to the order of the additions
            np.testing.assert_allclose(a[k], b[k], rtol=1e-6, atol=1e-9, err_msg=k)
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_frame_storage_refuses_ready_made_rows(hxlib):
    from isaac_amd.envs.configs import HectorCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.algo.ppo import PPO, ActorCritic
    cfg = HectorCfg(); cfg.env.num_envs = 32
    env = HectorFreeEnv(cfg)
    alg = PPO(ActorCritic(615, 1050, 10, [512, 256, 128], [768, 256, 128]), stream=env.stream)
    alg.init_storage(32, 4, [615], [1050], [10], obs_ld=616, priv_ld=1052, frames=env.frame_dims)
    with pytest.raises(RuntimeError, match="single-frame"):
        alg.act(env.get_observations(), env.get_privileged_observations())
    with pytest.raises(RuntimeError, match="storage_rows|do not exist"):
        alg.buffer(9, (4, 32, 616))
    alg.close(); env.close()


def test_deferred_critic_batching_does_not_change_the_values(hxlib, monkeypatch):
    """The critic runs beside the rollout in batches of HX_CRITIC_CHUNK slots (tile shape by batch size): however the slots are
    grouped -- one at a time, the default pairs, seven at a time (leaving a remainder for compute_returns), or all at the end --
    values, returns and advantages must come out the same (same fp32 k order in every tile shape: bitwise) -- and likewise with the
    critic's yield to the actor (HX_CRITIC_YIELD) switched off."""
    from isaac_amd import capi
    from isaac_amd.envs.configs import HectorCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.utils.helpers import set_seed
    from oracle.ppo import ActorCriticOracle
    N, T = 96, 20
    init = ActorCriticOracle.default_init(np.random.default_rng(4)).state_dict()
    res = []
    for chunk, yield_ in (("2", "1"), ("1", "1"), ("7", "1"), ("60", "1"), ("2", "0")):
        monkeypatch.setenv("HX_CRITIC_CHUNK", chunk)
        monkeypatch.setenv("HX_CRITIC_YIELD", yield_)      # the critic sleeping while actor workgroups are in flight is scheduling only
        cfg = HectorCfg(); cfg.env.num_envs = N; cfg.seed = set_seed(9)
        env = HectorFreeEnv(cfg)
        ac = ActorCritic(615, 1050, 10, [512, 256, 128], [768, 256, 128]); ac.load_state_dict(init)
        alg = PPO(ac, num_learning_epochs=1, num_mini_batches=4, gamma=0.994, lam=0.9, learning_rate=1e-5, schedule="adaptive",
                  desired_kl=0.01, stream=env.stream)
        alg.init_storage(N, T, [615], [1050], [10], obs_ld=616, priv_ld=1052)
        alg.rollout([env], T)
        alg.compute_returns(env.get_privileged_observations())
        res.append({k: alg.buffer(i, (T, N)).numpy().copy() for k, i in
                    (("values", capi.PPO_BUF_VALUES), ("returns", capi.PPO_BUF_RETURNS), ("adv", capi.PPO_BUF_ADVANTAGES))})
        env.close()
    for other in res[1:]:
        for k in res[0]:
            np.testing.assert_array_equal(res[0][k], other[k], err_msg=k)


def test_c_rollout_equals_stepwise_api(hxlib):
    """hx_rollout (zero-copy env -> storage, fused actor kernel, deferred critic) must fill the rollout storage exactly
    like the reference-style loop  act -> env.step -> process_env_step  driven through the per-call API."""
    from isaac_amd.envs.configs import HectorCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.utils.helpers import set_seed
    from oracle.ppo import ActorCriticOracle
    N, T = 128, 12
    init = ActorCriticOracle.default_init(np.random.default_rng(4)).state_dict()
    res = []
    for use_c in (False, True):
        cfg = HectorCfg(); cfg.env.num_envs = N; cfg.seed = set_seed(9)
        env = HectorFreeEnv(cfg)
        ac = ActorCritic(615, 1050, 10, [512, 256, 128], [768, 256, 128]); ac.load_state_dict(init)
        alg = PPO(ac, num_learning_epochs=1, num_mini_batches=4, gamma=0.994, lam=0.9, learning_rate=1e-5, schedule="adaptive",
                  desired_kl=0.01, stream=env.stream)
        alg.init_storage(N, T, [615], [1050], [10], obs_ld=616, priv_ld=1052)
        if use_c:
            alg.rollout([env], T)
            priv = env.get_privileged_observations()
        else:
            obs, priv = env.get_observations(), env.get_privileged_observations()
            for _ in range(T):
                a = alg.act(obs, priv)
                obs, priv, rew, done, infos = env.step(a)
                alg.process_env_step(rew, done, infos)
        alg.compute_returns(priv)
        res.append({k: alg.buffer(i, shp, dt).numpy() for k, i, shp, dt in
                    (("actions", 0, (T, N, 10), np.float32), ("values", 1, (T, N), np.float32), ("logp", 2, (T, N), np.float32),
                     ("rewards", 4, (T, N), np.float32), ("returns", 5, (T, N), np.float32), ("adv", 6, (T, N), np.float32))})
        res[-1]["final_obs"] = env.get_observations().numpy()
        env.close()
    a, b = res
    # identical kernels fed identical data -> identical bits; the only tolerance is for the actor, which the C path
    # evaluates with the fused 16x16x4-MFMA kernel instead of the three 32x32x2-MFMA GEMMs (different k order)
    np.testing.assert_allclose(a["actions"], b["actions"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["logp"], b["logp"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(a["values"], b["values"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["rewards"], b["rewards"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a["final_obs"], b["final_obs"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(a["adv"], b["adv"], rtol=0, atol=5e-3)


def test_hector_full_trains_through_the_registry(hxlib, tmp_path):
    """The 18-DoF sibling task (reference humanoid/envs/__init__.py:49, hector_w_arm_config.py) end to end: registry ->
    HectorFullFreeEnv (kernel instantiated with the arm chains) -> runner with its actor [768,512,128] / critic [768]*3."""
    from isaac_amd.envs import task_registry
    from isaac_amd.utils import get_args
    args = get_args(["--task=hector_full", "--headless", "--num_envs", "128", "--max_iterations", "2", "--seed", "3"])
    env, env_cfg = task_registry.make_env("hector_full", args=args)
    assert (env.num_obs, env.num_privileged_obs, env.num_actions) == (975, 1410, 18)
    runner, train_cfg = task_registry.make_alg_runner(env, name="hector_full", args=args, log_root=str(tmp_path))
    runner.learn(2, init_at_random_ep_len=True)
    rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    assert len(rows) == 2 and np.isfinite(rows[-1]["Loss/value_function"]) and np.isfinite(rows[-1]["Loss/surrogate"])
    sd = runner.alg.actor_critic.state_dict()
    assert sd["actor.0.weight"].shape == (768, 975) and sd["critic.0.weight"].shape == (768, 1410) and sd["std"].shape == (18,)
    obs = env.get_observations().numpy()
    assert obs.shape == (128, 975) and np.all(np.isfinite(obs))
    act = runner.get_inference_policy()(env.get_observations()).numpy()
    assert act.shape == (128, 18) and np.all(np.isfinite(act))
    root, q, qd = env.get_state()
    assert q.shape == (128, 18) and np.all(np.isfinite(q)) and np.all(np.isfinite(root))
    env.close()


def test_humanoid_ppo_trains_through_the_registry(hxlib, tmp_path):
    """The XBot-L task (reference humanoid/envs/__init__.py:47, humanoid_config.py / humanoid_env.py) end to end: registry ->
    XBotLFreeEnv (kernel instantiated with rotated joint frames and 6-joint legs) -> runner; 47 x 15 observations, a 73 x 3
    privileged row, 12 actions, checkpoint with the reference's tensor shapes."""
    from isaac_amd.envs import task_registry
    from isaac_amd.utils import get_args
    args = get_args(["--task=humanoid_ppo", "--headless", "--num_envs", "128", "--max_iterations", "2", "--seed", "3"])
    env, env_cfg = task_registry.make_env("humanoid_ppo", args=args)
    assert (env.num_obs, env.num_privileged_obs, env.num_actions) == (705, 219, 12)
    runner, train_cfg = task_registry.make_alg_runner(env, name="humanoid_ppo", args=args, log_root=str(tmp_path))
    runner.learn(2, init_at_random_ep_len=True)
    rows = [json.loads(l) for l in open(os.path.join(runner.log_dir, "scalars.jsonl"))]
    assert len(rows) == 2 and np.isfinite(rows[-1]["Loss/value_function"]) and np.isfinite(rows[-1]["Loss/surrogate"])
    assert "Episode/rew_joint_pos" in rows[-1] and "Episode/rew_track_vel_hard" in rows[-1]          # this task's reward set
    sd = runner.alg.actor_critic.state_dict()
    assert sd["actor.0.weight"].shape == (512, 705) and sd["critic.0.weight"].shape == (768, 219) and sd["std"].shape == (12,)
    obs, priv = env.get_observations().numpy(), env.get_privileged_observations().numpy()
    assert obs.shape == (128, 705) and priv.shape == (128, 219) and np.all(np.isfinite(obs)) and np.all(np.abs(obs) <= 18.0)
    root, q, qd = env.get_state()
    assert q.shape == (128, 12) and np.all(np.isfinite(q)) and 0.6 < np.median(root[:, 2] - env.env_origins[:, 2]) < 1.1   # standing height
    env.close()


@pytest.mark.parametrize("task", ["hector", "humanoid_ppo"])
def test_pause_words_are_zero_after_every_rollout(hxlib, task):
    """The background critic sleeps while a device count is up: the fused actor raises it per workgroup and lowers it again, and the
    simulator's stacking launch raises it once for itself with the NEXT fused-actor launch taking that 1 back (hx_sim_set_pause_word).
    A count that is left up costs every critic workgroup its whole sleep budget (round 4: with the layer-by-layer rollout actor of
    humanoid_ppo / hector_full nobody took the stacking launch's 1 back, and their iterations were 8 % slower until the hand-over was
    restricted to the fused actor).  After a rollout -- whole, in pieces, followed by an update -- both words must read 0."""
    import ctypes
    from isaac_amd.envs.configs import HectorCfg, XBotLCfg
    from isaac_amd.envs.hector_env import HectorFreeEnv, XBotLFreeEnv
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.utils.helpers import set_seed
    N, T = 128, 8
    cfg_cls, env_cls = (HectorCfg, HectorFreeEnv) if task == "hector" else (XBotLCfg, XBotLFreeEnv)
    cfg = cfg_cls(); cfg.env.num_envs = N; cfg.seed = set_seed(2)
    if task != "hector":
        cfg.terrain.mesh_type = "plane"
    env = env_cls(cfg)
    dims = ([512, 256, 128], [768, 256, 128]) if task == "hector" else ([256, 128, 64], [256, 128, 64])
    alg = PPO(ActorCritic(env.num_obs, env.num_privileged_obs, env.num_actions, *dims), num_learning_epochs=1, num_mini_batches=2, stream=env.stream)
    alg.init_storage(N, T, [env.num_obs], [env.num_privileged_obs], [env.num_actions], obs_ld=env.obs_ld, priv_ld=env.priv_ld)
    words = (ctypes.c_int32 * 2)()

    def check(when):
        rc = alg._L.hx_ppo_pause_words(alg._h, words)
        assert rc == 0 and (words[0], words[1]) == (0, 0), f"{when}: pause words {words[0]}, {words[1]}"

    check("before any rollout")
    for pieces in ((T,), (3, T - 3), (1,) * T):
        for n in pieces:
            alg.rollout([env], n)
            check(f"after a rollout piece of {n} steps")
        alg.compute_returns(env.get_privileged_observations())
        alg.update()
        check("after the update")
    alg.close(); env.close()
