"""VecEnv protocol (reference humanoid/algo/vec_env.py:37-61): what the runner may touch on an env."""
from abc import ABC, abstractmethod


class VecEnv(ABC):
    num_envs: int
    num_obs: int
    num_privileged_obs: int
    num_actions: int
    max_episode_length: int

    @abstractmethod
    def step(self, actions):
        """-> (obs, privileged_obs, rewards, dones, infos)"""

    @abstractmethod
    def reset(self):
        """-> (obs, privileged_obs)"""

    @abstractmethod
    def get_observations(self):
        pass

    @abstractmethod
    def get_privileged_observations(self):
        pass
