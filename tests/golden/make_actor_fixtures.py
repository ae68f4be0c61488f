#!/usr/bin/env python3
"""Container-only: extract the weights of the seven hector actors the reference SHIPS (humanoid/locomotion_net*.onnx and
locomotion_net.onnx at its root -- ONNX exports of policies trained against PhysX, reference play.py:89-98) into
tests/golden/actors/<name>.npz.  Data only: four (weight, bias) pairs of a 615-512-256-128-10 ELU MLP per file, fp32,
bit-exact; no reference source text.  They are the only PhysX-derived artefacts available offline, so the physics
fidelity test (tests/test_gpu_fidelity.py) rolls them on the HIP simulator.
usage: python tests/golden/make_actor_fixtures.py"""
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from isaac_amd.utils import onnx_io  # noqa: E402

REF = "/root/reference"


def main():
    out = os.path.join(HERE, "actors")
    os.makedirs(out, exist_ok=True)
    files = sorted(glob.glob(os.path.join(REF, "humanoid", "locomotion_net*.onnx"))) + [os.path.join(REF, "locomotion_net.onnx")]
    index = {}
    for f in files:
        name = os.path.basename(f)[:-5] + ("_root" if os.path.dirname(f) == REF else "")
        layers = onnx_io.load_actor(f)
        arrs = {}
        for i, (W, b) in enumerate(layers):
            arrs[f"{2 * i}.weight"] = np.ascontiguousarray(W, np.float32)
            arrs[f"{2 * i}.bias"] = np.ascontiguousarray(b, np.float32)
        np.savez_compressed(os.path.join(out, name + ".npz"), **arrs)
        index[name] = dict(source=os.path.relpath(f, REF), md5=hashlib.md5(open(f, "rb").read()).hexdigest(),
                           shapes=[list(W.shape) for W, _ in layers])
        print(name, index[name]["shapes"])
    with open(os.path.join(out, "index.json"), "w") as fh:
        json.dump(index, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
