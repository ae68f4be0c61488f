from .helpers import class_to_dict, export_policy_as_jit, export_policy_as_onnx, get_args, get_load_path, set_seed, update_class_from_dict  # noqa: F401
from .task_registry import task_registry  # noqa: F401
from .logger import Logger  # noqa: F401
