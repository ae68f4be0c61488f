#!/usr/bin/env python3
"""Where do robots fall on the default tile map (mesh_type 'trimesh', 20 x 20 tiles, random difficulty rows as in the
reference's own training)?  Rolls a policy -- the PhysX-trained actor the reference ships, or a checkpoint trained here --
with the play protocol (fixed command vx = 0.5 m/s, 10 s, 4096 robots) and prints survival and falls per robot per 10 s by
tile kind x difficulty tercile, for the build's default and for ablations of the terrain / contact model:

  default        trimesh semantics of the reference (walls from slope_treshold, cliff cells flattened), PhysX contact inputs
  ramps          no walls at all (mesh_type 'heightfield' semantics: wall_height = 0)
  no_flatten     walls push sideways, but cliff cells keep their ramp surface
  no_wallpush    cliff cells flattened, but no sideways wall contact
  plain_contact  max_depenetration_velocity = 0, contact_offset = 0 (the round-2 spring-damper)
  no_depen_cap / no_offset   one of the two at a time

usage: python tools/falls_by_tile.py [--policy physx|<model_N.pt>] [--variants default,ramps,...] [--push on|off] [--steps 1000]
GPU box only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd.utils.actor_eval import load_actor_checkpoint, load_actor_npz, roll_actor  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--policy", default="physx")
ap.add_argument("--variants", default="default,ramps,no_flatten,no_wallpush,plain_contact")
ap.add_argument("--push", default="off", choices=["on", "off"])
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--cmd", default="0.5", help="fixed forward command in m/s (the play protocol), or 'env' for the env's own resampled commands (training conditions)")
ap.add_argument("--json", default=None, help="also append one JSON line per variant to this file")
args = ap.parse_args()
sd = load_actor_npz(os.path.join(ROOT, "tests", "golden", "actors", "locomotion_net.npz")) if args.policy == "physx" else load_actor_checkpoint(args.policy)

VARIANTS = {
    "default": dict(),
    "ramps": dict(edit=lambda c: setattr(c.terrain, "slope_treshold", None)),
    "no_flatten": dict(flags=1),
    "no_wallpush": dict(flags=2),
    "plain_contact": dict(phys=dict(max_depenetration_velocity=0.0, contact_offset=0.0)),
    "no_depen_cap": dict(phys=dict(max_depenetration_velocity=0.0)),
    "no_offset": dict(phys=dict(contact_offset=0.0)),
}
for name in args.variants.split(","):
    v = VARIANTS[name]

    def edit(cfg, v=v):
        cfg.domain_rand.push_robots = args.push == "on"
        if "edit" in v:
            v["edit"](cfg)
    r = roll_actor(sd, num_envs=args.envs, steps=args.steps, mesh_type="trimesh", cfg_edit=edit, phys=v.get("phys"),
                   command=None if args.cmd == "env" else (float(args.cmd), 0.0, 0.0, 0.0),
                   terrain_flags=v.get("flags", 0), by_tile=True)
    print(f"== policy {os.path.basename(args.policy)}  variant {name}  pushes {args.push}  command {args.cmd}: survival {r['survival']:.3f}  "
          f"falls/robot/10s {r['falls_per_robot_10s']:.2f}  median first fall {r['median_first_fall']:.0f} steps  vx of robots still up {r['mean_vx']:.3f}")
    print("kind          robots  survival  falls/robot/10s |  survival by difficulty tercile (easy, mid, hard) | falls/robot/10s by tercile")
    tiles = r["tiles"]
    for kind in dict.fromkeys(t["kind"] for t in tiles):
        rows = {t["tercile"]: t for t in tiles if t["kind"] == kind}
        a = rows["all"]
        g = lambda k, f: ("%6.2f" % rows[k][f]) if k in rows else "     -"
        print(f"{kind:12s} {a['robots']:6d}   {a['survival']:6.3f}   {a['falls_per_robot_10s']:8.2f}       |   "
              + "  ".join(g(k, "survival") for k in ("easy", "mid", "hard")) + "        |   " + "  ".join(g(k, "falls_per_robot_10s") for k in ("easy", "mid", "hard")))
    sys.stdout.flush()
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(dict(policy=os.path.basename(args.policy), variant=name, push=args.push, **r)) + "\n")
