"""ONNX actor files without the onnx package: writer/reader round trip, the graph the reference's
`torch.onnx.export(actor, ...)` produces (play.py:89-98), rejection of anything that is not a Gemm/Elu chain, and --
when the reference checkout is present -- the seven actors it ships."""
import glob
import os

import numpy as np
import pytest

from isaac_amd.utils import onnx_io as O
from isaac_amd.utils.logger import Logger


def _layers(rng, dims):
    return [(rng.standard_normal((o, i)).astype(np.float32), rng.standard_normal(o).astype(np.float32))
            for i, o in zip(dims[:-1], dims[1:])]


def test_round_trip_and_graph_shape(tmp_path):
    rng = np.random.default_rng(0)
    layers = _layers(rng, [615, 512, 256, 128, 10])
    path = O.save_actor(str(tmp_path / "actor.onnx"), layers)
    m = O.load_model(path)
    assert (m["ir_version"], m["opset"]) == (6, 11)
    assert [n["op"] for n in m["nodes"]] == ["Gemm", "Elu", "Gemm", "Elu", "Gemm", "Elu", "Gemm"]
    assert m["nodes"][0]["attrs"] == {"alpha": 1.0, "beta": 1.0, "transB": 1}
    assert m["nodes"][0]["inputs"] == ["obs", "0.weight", "0.bias"] and m["nodes"][-1]["outputs"] == ["action"]
    assert sorted(m["initializers"]) == sorted(f"{2 * i}.{k}" for i in range(4) for k in ("weight", "bias"))
    assert m["inputs"] == ["obs"] and m["outputs"] == ["action"]
    back = O.load_actor(path)
    for (W, b), (W2, b2) in zip(layers, back):
        assert np.array_equal(W, W2) and np.array_equal(b, b2)
    sd = O.actor_state_dict(path)
    assert list(sd) == [f"actor.{2 * i}.{k}" for i in range(4) for k in ("weight", "bias")]


def test_forward_matches_oracle_mlp(tmp_path):
    from oracle.ppo import MLP
    rng = np.random.default_rng(1)
    layers = _layers(rng, [615, 64, 32, 10])
    x = rng.standard_normal((5, 615)).astype(np.float32)
    mlp = MLP([W for W, _ in layers], [b for _, b in layers])
    np.testing.assert_allclose(O.mlp_forward(O.load_actor(O.save_actor(str(tmp_path / "a.onnx"), layers)), x),
                               mlp.forward(x)[0] if isinstance(mlp.forward(x), tuple) else mlp.forward(x), rtol=2e-5, atol=2e-5)


def test_rejects_non_mlp_graphs(tmp_path):
    rng = np.random.default_rng(2)
    layers = _layers(rng, [8, 4, 2])
    path = str(tmp_path / "a.onnx")
    O.save_actor(path, layers)
    data = open(path, "rb").read()
    bad = str(tmp_path / "bad.onnx")
    open(bad, "wb").write(data.replace(b"\x22\x03Elu", b"\x22\x03Exp"))      # op_type of the activation node
    with pytest.raises(ValueError, match="unexpected op Exp"):
        O.load_actor(bad)
    mismatched = [layers[0], (rng.standard_normal((2, 5)).astype(np.float32), layers[1][1])]
    with pytest.raises(ValueError, match="do not chain"):
        O.load_actor(O.save_actor(str(tmp_path / "m.onnx"), mismatched))


REF_ONNX = sorted(glob.glob("/root/reference/humanoid/*.onnx"))


@pytest.mark.skipif(not REF_ONNX, reason="reference checkout not present (files are not copied into this repository)")
def test_parses_the_actors_the_reference_ships():
    assert len(REF_ONNX) >= 6
    for f in REF_ONNX:
        m = O.load_model(f)
        assert (m["ir_version"], m["opset"], m["producer"]) == (6, 11, "pytorch")
        layers = O.load_actor(f)
        assert [W.shape for W, _ in layers] == [(512, 615), (256, 512), (128, 256), (10, 128)]
        assert m["inputs"] == ["obs"] and m["outputs"] == ["action"]
        out = O.mlp_forward(layers, np.zeros((1, 615), np.float32))
        assert out.shape == (1, 10) and np.all(np.isfinite(out)) and np.abs(out).max() < 20.0


def test_logger_traces(tmp_path):
    lg = Logger(0.01)
    for i in range(5):
        lg.log_states({"dof_pos": 0.1 * i, "command_x": 0.5, "contact_forces_z": np.array([1.0 * i, 2.0])})
    lg.log_rewards({"rew_tracking_lin_vel": 0.25, "terrain_level": 3.0}, 2)
    out = lg.plot_states(str(tmp_path / "tr" / "play_states"))
    d = np.load(out)
    assert d["dof_pos"].shape == (5,) and d["contact_forces_z"].shape == (5, 2)
    np.testing.assert_allclose(d["time"], np.arange(5) * 0.01)
    assert open(str(tmp_path / "tr" / "play_states.csv")).readline().strip().split(",") == ["dof_pos", "command_x", "time"]
    assert lg.num_episodes == 2 and lg.rew_log["rew_tracking_lin_vel"] == [0.5] and "terrain_level" not in lg.rew_log
