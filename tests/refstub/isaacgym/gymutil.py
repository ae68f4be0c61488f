def parse_device_str(s):
    parts = s.split(":")
    return parts[0], int(parts[1]) if len(parts) > 1 else 0


def parse_sim_config(cfg, sim_params):
    for k, v in cfg.items():
        if k == "physx":
            for kk, vv in v.items():
                setattr(sim_params.physx, kk, vv)
        else:
            setattr(sim_params, k, v)
