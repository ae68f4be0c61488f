#!/usr/bin/env python3
"""tests/golden/configs.json = class_to_dict(HectorCfg()), class_to_dict(HectorCfgPPO()) (and the hector_full pair) computed by the
REFERENCE's own classes and helper (humanoid/envs/custom/hector_config.py, humanoid/utils/helpers.py:43-58).
Pins every constant of the benchmark and the alphabetical reward order.  Run in this container only."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.refstub import loader  # noqa: E402

env_mod, cfg_mod, helpers = loader.load_env()
out = {"HectorCfg": helpers.class_to_dict(cfg_mod.HectorCfg()), "HectorCfgPPO": helpers.class_to_dict(cfg_mod.HectorCfgPPO())}
_, cfg_full, _ = loader.load_env("hector_full")          # sibling task (reference hector_w_arm_config.py), SURVEY 8f-4
out["HectorFullCfg"] = helpers.class_to_dict(cfg_full.HectorFullCfg())
out["HectorFullCfgPPO"] = helpers.class_to_dict(cfg_full.HectorFullCfgPPO())
_, cfg_x, _ = loader.load_env("humanoid_ppo")         # sibling task (reference humanoid_config.py): configs only so far
out["XBotLCfg"] = helpers.class_to_dict(cfg_x.XBotLCfg())
out["XBotLCfgPPO"] = helpers.class_to_dict(cfg_x.XBotLCfgPPO())
with open(os.path.join(HERE, "configs.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print("reward order:", [k for k, v in out["HectorCfg"]["rewards"]["scales"].items() if v != 0])
