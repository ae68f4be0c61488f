from .ppo import PPO, ActorCritic  # noqa: F401
from .on_policy_runner import OnPolicyRunner  # noqa: F401
from .vec_env import VecEnv  # noqa: F401
