"""Config helpers shared by the env and the CLI helpers (leaf module: imports nothing from the package)."""


def class_to_dict(obj):
    """reference humanoid/utils/helpers.py:43-58 -- dir() order == alphabetical (decides reward order)."""
    if not hasattr(obj, "__dict__"):
        return obj
    out = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        out[key] = [class_to_dict(i) for i in val] if isinstance(val, list) else class_to_dict(val)
    return out
