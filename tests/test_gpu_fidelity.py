"""GPU: physics fidelity evidence for SURVEY row a4 (`gym.simulate`, parity unpinned -- PhysX is unavailable).

The only PhysX-derived artefacts in the reference are the seven hector actors it ships as ONNX
(humanoid/locomotion_net*.onnx, locomotion_net.onnx; weights extracted bit-exactly to tests/golden/actors/ by
tests/golden/make_actor_fixtures.py).  A policy trained on one simulator walking zero-shot on another is the strongest
statement available without the reference binary, so each actor is rolled on the HIP simulator: 4096 robots, ground
plane, the default HectorCfg (friction / payload randomisation, observation and action noise, pushes), the play script's
fixed command vx = 0.5 m/s (play.py:136-140), 10 s.  Thresholds are written in THRESHOLDS below and in DESIGN.md section 4.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ACTORS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "actors")

# actor -> (min survival over 10 s, (lo, hi) band of the mean forward speed of surviving robots in m/s) or "falls"
# measured on the round-2 build (profiles/r02_a_actor_fidelity.txt): locomotion_net 0.626 / 0.347 m/s with the default
# pushes (velocity overwrites of +-0.3 m/s / +-0.4 rad/s every 4 s), 1.000 / 0.361 m/s without them
# With pushes 0.63 of the robots survive; every fall follows a push (1.000 without them).  A regression guard around the
# measurement, not a fidelity bar: PhysX resolves the velocity overwrite of a push through rigid contacts within the same
# step, a penalty contact cannot; the number is printed by the test.
THRESHOLDS = {
    "locomotion_net": (0.58, (0.30, 0.40)),
}
THRESHOLDS_NO_PUSH = {                 # play.py's own protocol: domain_rand.push_robots = False (play.py:57)
    "locomotion_net": (0.98, (0.32, 0.40)),                  # measured 0.360-0.363 m/s on the 0.5 m/s command (why 0.72)
    "locomotion_net_root": (0.95, None),                     # this actor marches on the spot in this simulator (vx 0.03-0.04): no speed clause
    "locomotion_net_newkp_passive": "falls", "locomotion_net_kp_10_test": "falls", "locomotion_net_hop_tst": "falls",
    "locomotion_net_bound_test": "falls", "locomotion_net_active_ankle_new_test": "falls",   # trained for other gain sets / gaits
}


def test_actor_fixtures_are_complete():
    idx = json.load(open(os.path.join(ACTORS, "index.json")))
    assert len(idx) == 7
    for name, meta in idx.items():
        d = np.load(os.path.join(ACTORS, name + ".npz"))
        assert [list(d[f"{2 * i}.weight"].shape) for i in range(4)] == meta["shapes"] == [[512, 615], [256, 512], [128, 256], [10, 128]]


def _no_push(cfg):
    cfg.domain_rand.push_robots = False


@pytest.mark.parametrize("name,push", [(n, True) for n in sorted(THRESHOLDS)] + [(n, False) for n in sorted(THRESHOLDS_NO_PUSH)])
def test_shipped_actor_walks(hxlib, name, push):
    from isaac_amd.utils.actor_eval import load_actor_npz, roll_actor
    r = roll_actor(load_actor_npz(os.path.join(ACTORS, name + ".npz")), num_envs=4096, steps=1000, cfg_edit=None if push else _no_push)
    print(name, "pushes" if push else "no pushes", json.dumps(r))
    want = (THRESHOLDS if push else THRESHOLDS_NO_PUSH)[name]
    if want == "falls":
        assert r["survival"] < 0.10, r
    else:
        assert r["survival"] >= want[0], r
        if want[1] is not None:
            assert want[1][0] <= r["mean_vx"] <= want[1][1], r


def test_shipped_actor_on_the_default_tile_map(hxlib):
    """The same actor on the reference's DEFAULT terrain (mesh_type 'trimesh', hector_config.py:45: 20 x 20 tiles, every robot on
    a random difficulty row as in the reference's own training; play.py's protocol otherwise: vx = 0.5 m/s, pushes off, 10 s).
    What can be ARGUED without the PhysX binary, and is asserted:
      * a flat tile is the ground plane: survival there must equal the plane protocol's (>= 0.98, test above);
      * the actor is blind (measure_heights = False, hector_config.py:49) and its gait lifts the swing foot by the reward's
        target_feet_height = 0.06 m (hector_config.py:151): obstacles taller than that trip it in ANY simulator.  The generator
        scales obstacle height with the tile's difficulty d (utils/terrain.py:213-215: steps 0.2 d, roughness 0.14 d, slopes
        0.45 d), so survival must fall with the difficulty tercile on every non-flat kind, and the hard tercile (steps
        >= 0.13 m, slopes >= 0.3 with friction as low as 0.35) must be lost almost entirely;
      * walls vs ramps, cliff-cell flattening and the PhysX contact inputs must NOT matter for this number (they move it by
        < 0.01, profiles/r03_a_falls_by_tile.txt) -- if they start to, the terrain contact changed.
    What cannot be argued is the level on the easy / mid terciles (0.5-0.77 / 0.0-0.26 measured): whether PhysX keeps more of
    these robots up is exactly the unpinned part of row a4.  The overall survival is therefore asserted as a regression
    band around the measurement (0.288 +- 0.04), not as a fidelity bar; the number is printed."""
    from isaac_amd.utils.actor_eval import load_actor_npz, roll_actor
    sd = load_actor_npz(os.path.join(ACTORS, "locomotion_net.npz"))
    r = roll_actor(sd, num_envs=4096, steps=1000, mesh_type="trimesh", cfg_edit=_no_push, by_tile=True)
    tiles = {(t["kind"], t["tercile"]): t for t in r["tiles"]}
    print("locomotion_net on the default tile map: survival %.3f, falls/robot/10s %.2f" % (r["survival"], r["falls_per_robot_10s"]))
    for (kind, terc), t in sorted(tiles.items()):
        print("  %-12s %-5s robots %4d survival %.3f" % (kind, terc, t["robots"], t["survival"]))
    assert tiles[("flat", "all")]["survival"] >= 0.98, tiles[("flat", "all")]
    for kind in ("rough", "slope up", "slope down", "stairs up", "stairs down"):
        e, m, h = (tiles[(kind, k)]["survival"] for k in ("easy", "mid", "hard"))
        assert e + 0.05 >= m and m + 0.05 >= h, (kind, e, m, h)
        assert h <= 0.05, (kind, h)
        assert e >= 0.35, (kind, e)                     # regression guard: measured 0.49 - 0.74
    assert 0.25 <= r["survival"] <= 0.33, r["survival"]   # regression band around 0.288 (see docstring)
    # ablations: ramps instead of walls, and the plain spring-damper, move the number by less than 0.02
    r2 = roll_actor(sd, num_envs=4096, steps=1000, mesh_type="trimesh", phys=dict(max_depenetration_velocity=0.0, contact_offset=0.0),
                    cfg_edit=lambda c: (_no_push(c), setattr(c.terrain, "slope_treshold", None)))
    print("  ramps + plain contact: survival %.3f" % r2["survival"])
    assert abs(r2["survival"] - r["survival"]) < 0.03, (r["survival"], r2["survival"])
