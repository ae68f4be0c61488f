"""isaac_amd -- MI355X-native hot path of DRCL-USC/isaac: vectorised hector env step + PPO learner as
hand-written HIP kernels behind a C ABI (include/hx_sim.h, include/hx_ppo.h), with the reference's
config / registry / runner surface on top.  See DESIGN.md."""
import os

LEGGED_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
LEGGED_GYM_ENVS_DIR = os.path.join(LEGGED_GYM_ROOT_DIR, "isaac_amd", "envs")
__version__ = "0.1.0"
