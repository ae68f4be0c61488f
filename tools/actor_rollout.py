#!/usr/bin/env python3
"""GPU: roll the actors the reference ships (tests/golden/actors/*.npz, trained against PhysX) on the HIP simulator and
print one JSON line per (actor, variant): survival, forward speed, height.  Variants override the contact-model constants
(`--phys contact_kn=8e4,friction_veps=0.005`), several --phys options = several variants.
usage: python tools/actor_rollout.py [--actors a,b] [--envs 4096] [--steps 1000] [--terrain plane] [--phys k=v,...]... [--nodr]"""
import argparse
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd.utils.actor_eval import load_actor_npz, roll_actor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--actors", default="all")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--terrain", default="plane")
ap.add_argument("--vx", type=float, default=0.5)
ap.add_argument("--phys", action="append", default=[])
ap.add_argument("--nodr", action="store_true", help="no friction / mass randomisation, no pushes, no noise")
ap.add_argument("--off", default="", help="comma list of DR components to switch off: friction,mass,push,noise,actnoise")
ap.add_argument("--diag", action="store_true")
args = ap.parse_args()
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "actors")
names = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(root, "*.npz"))) if args.actors == "all" else args.actors.split(",")
variants = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in v.split(",") if kv) for v in (args.phys or [""])]


def edit(cfg):
    off = set(args.off.split(",")) if args.off else set()
    if "friction" in off: cfg.domain_rand.randomize_friction = False
    if "mass" in off: cfg.domain_rand.randomize_base_mass = False
    if "push" in off: cfg.domain_rand.push_robots = False
    if "noise" in off: cfg.noise.add_noise = False
    if "actnoise" in off: cfg.domain_rand.action_noise = 0.0
    if args.nodr:
        cfg.domain_rand.randomize_friction = False
        cfg.domain_rand.randomize_base_mass = False
        cfg.domain_rand.push_robots = False
        cfg.noise.add_noise = False
        cfg.domain_rand.action_noise = 0.0


for v in variants:
    for nm in names:
        r = roll_actor(load_actor_npz(os.path.join(root, nm + ".npz")), num_envs=args.envs, steps=args.steps,
                       command=(args.vx, 0.0, 0.0, 0.0), mesh_type=args.terrain, cfg_edit=edit, phys=v, diagnostics=args.diag)
        r.update(actor=nm, phys=v, nodr=args.nodr, off=args.off, terrain=args.terrain)
        print(json.dumps(r), flush=True)
