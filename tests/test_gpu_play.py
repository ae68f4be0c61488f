"""GPU: the headless play script (reference humanoid/scripts/play.py) end to end with an actor loaded from ONNX:
environment overrides, policy export (TorchScript + ONNX), fixed command, state traces."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_play_with_onnx_actor(hxlib, tmp_path):
    from isaac_amd.scripts.play import play
    from isaac_amd.utils import get_args, onnx_io

    rng = np.random.default_rng(3)
    dims = [615, 512, 256, 128, 10]
    layers = [((rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32), (0.01 * rng.standard_normal(o)).astype(np.float32))
              for i, o in zip(dims[:-1], dims[1:])]
    src = onnx_io.save_actor(str(tmp_path / "trained.onnx"), layers)
    out = str(tmp_path / "out")
    args = get_args(["--task=hector", "--headless", "--onnx", src, "--play_steps", "40", "--play_out", out])
    logger = play(args)

    d = np.load(os.path.join(out, "play_states.npz"))
    for k in ("dof_pos_target", "dof_pos", "dof_vel", "dof_torque", "command_x", "command_y", "command_yaw", "base_vel_x",
              "base_vel_y", "base_vel_z", "base_vel_yaw", "contact_forces_z", "time"):
        assert k in d, k
    assert d["dof_pos"].shape == (40,) and d["contact_forces_z"].shape == (40, 2)
    np.testing.assert_allclose(d["command_x"], 0.5)              # FIX_COMMAND (play.py:136-140)
    np.testing.assert_allclose(d["command_y"], 0.0)
    assert np.all(np.isfinite(d["base_vel_x"])) and np.abs(d["dof_torque"]).max() <= 28.475 + 1e-3
    assert d["contact_forces_z"].max() > 10.0                     # the logged robot stood on the tile map

    # exported policies carry exactly the loaded actor
    exported = onnx_io.load_actor(os.path.join(out, "policies", "locomotion_net.onnx"))
    for (W, b), (W2, b2) in zip(layers, exported):
        assert np.array_equal(W, W2) and np.array_equal(b, b2)
    import torch
    jit = torch.jit.load(os.path.join(out, "policies", "policy_1.pt"))
    x = rng.standard_normal((4, 615)).astype(np.float32)
    np.testing.assert_allclose(jit(torch.from_numpy(x)).detach().numpy(), onnx_io.mlp_forward(layers, x), rtol=1e-4, atol=1e-4)
    assert logger.state_log["dof_pos"]


def test_play_hector_full(hxlib, tmp_path):
    """The same script on the 18-DoF sibling task: overrides, a [975 -> 768 -> 512 -> 128 -> 18] actor from ONNX, traces."""
    from isaac_amd.scripts.play import play
    from isaac_amd.utils import get_args, onnx_io
    rng = np.random.default_rng(4)
    dims = [975, 768, 512, 128, 18]
    layers = [((rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32), (0.01 * rng.standard_normal(o)).astype(np.float32))
              for i, o in zip(dims[:-1], dims[1:])]
    src = onnx_io.save_actor(str(tmp_path / "full.onnx"), layers)
    out = str(tmp_path / "out_full")
    logger = play(get_args(["--task=hector_full", "--headless", "--onnx", src, "--play_steps", "30", "--play_out", out]))
    d = np.load(os.path.join(out, "play_states.npz"))
    assert d["dof_pos"].shape == (30,) and np.all(np.isfinite(d["base_vel_x"])) and np.all(np.isfinite(d["dof_torque"]))
    np.testing.assert_allclose(d["command_x"], 0.5)
    exported = onnx_io.load_actor(os.path.join(out, "policies", "locomotion_net.onnx"))
    for (W, b), (W2, b2) in zip(layers, exported):
        assert np.array_equal(W, W2) and np.array_equal(b, b2)
    assert logger.state_log["dof_pos"]


def test_device_inference_equals_onnx_chain(hxlib, tmp_path):
    """ActorCritic.load_actor_from_onnx + act_inference (hx_ppo_inference) against the numpy chain."""
    from isaac_amd.algo.ppo import PPO, ActorCritic
    from isaac_amd.utils import onnx_io
    rng = np.random.default_rng(5)
    dims = [615, 512, 256, 128, 10]
    layers = [((rng.standard_normal((o, i)) / np.sqrt(i)).astype(np.float32), (0.1 * rng.standard_normal(o)).astype(np.float32))
              for i, o in zip(dims[:-1], dims[1:])]
    path = onnx_io.save_actor(str(tmp_path / "a.onnx"), layers)
    ac = ActorCritic(615, 1050, 10, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[768, 256, 128])
    alg = PPO(ac)
    alg.init_storage(32, 4, [615], [1050], [10])
    critic_before = {k: v.copy() for k, v in ac.state_dict().items() if k.startswith("critic")}
    ac.load_actor_from_onnx(path)
    x = rng.standard_normal((32, 615)).astype(np.float32)
    np.testing.assert_allclose(ac.act_inference(x).numpy(), onnx_io.mlp_forward(layers, x), rtol=0, atol=5e-5)
    for k, v in ac.state_dict().items():
        if k.startswith("critic"):
            assert np.array_equal(v, critic_before[k])
    bad = onnx_io.save_actor(str(tmp_path / "b.onnx"), layers[:1] + [(layers[1][0][:, :100].copy(), layers[1][1])])
    with pytest.raises(ValueError):
        ac.load_actor_from_onnx(bad)
    alg.close()
