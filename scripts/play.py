"""Drop-in entry point: `python scripts/play.py --task=hector --load_run <run>` (or `--onnx <actor.onnx>`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd.scripts.play import play  # noqa: E402
from isaac_amd.utils import get_args  # noqa: E402

if __name__ == "__main__":
    play(get_args())
