"""Drop-in entry point: `python scripts/train.py --task=hector --run_name v1 --headless --num_envs 4096`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd.scripts.train import train  # noqa: E402
from isaac_amd.utils import get_args  # noqa: E402

if __name__ == "__main__":
    train(get_args())
