"""State / reward logger of the play script (reference humanoid/utils/logger.py): same methods and keys.
`plot_states` writes the traces to disk (npz + CSV) and draws the reference's 3x3 figure only when matplotlib is
importable and a path is given -- this build is headless."""
import os
from collections import defaultdict

import numpy as np


class Logger:
    def __init__(self, dt):
        self.state_log = defaultdict(list)
        self.rew_log = defaultdict(list)
        self.dt = dt
        self.num_episodes = 0

    def log_state(self, key, value):
        self.state_log[key].append(value)

    def log_states(self, d):
        for key, value in d.items():
            self.log_state(key, value)

    def log_rewards(self, d, num_episodes):
        for key, value in d.items():
            if "rew" in key:
                self.rew_log[key].append(float(value) * num_episodes)
        self.num_episodes += num_episodes

    def reset(self):
        self.state_log.clear()
        self.rew_log.clear()

    def print_rewards(self):
        print("Average rewards per second:")
        for key, values in self.rew_log.items():
            mean = np.sum(np.array(values)) / max(self.num_episodes, 1)
            print(f" - {key}: {mean}")
        print(f"Total number of episodes: {self.num_episodes}")

    def save_states(self, path):
        """<path>.npz with one array per logged key (+ `time`), and <path>.csv for scalar keys."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        arrays = {k: np.asarray(v) for k, v in self.state_log.items()}
        n = max((len(v) for v in arrays.values()), default=0)
        arrays["time"] = np.arange(n) * self.dt
        np.savez(path + ".npz", **arrays)
        scalar = [k for k, v in arrays.items() if v.ndim == 1 and len(v) == n]
        with open(path + ".csv", "w") as f:
            f.write(",".join(scalar) + "\n")
            for i in range(n):
                f.write(",".join(repr(float(arrays[k][i])) for k in scalar) + "\n")
        return path + ".npz"

    def plot_states(self, path=None):
        if path is None:
            return None
        out = self.save_states(path)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return out
        log = self.state_log
        fig, axs = plt.subplots(3, 3, figsize=(15, 10))
        t = np.arange(len(log["dof_pos"])) * self.dt
        panels = [("dof_pos", "dof_pos_target"), ("dof_vel",), ("base_vel_x", "command_x"), ("base_vel_y", "command_y"),
                  ("base_vel_yaw", "command_yaw"), ("base_vel_z",), ("contact_forces_z",), ("dof_torque",)]
        for ax, keys in zip(axs.flat, panels):
            for k in keys:
                if log.get(k):
                    ax.plot(t[:len(log[k])], np.asarray(log[k]), label=k)
            ax.legend(fontsize=7)
        fig.savefig(path + ".png", dpi=80)
        plt.close(fig)
        return out
