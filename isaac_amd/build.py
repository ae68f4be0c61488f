"""Build libhx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("hx_sim.hip", "hx_ppo.hip")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("hx_dyn.h", "hx_gemm.h", "hx_common.h", "hx_model_data.h")] + \
      [os.path.join(os.path.dirname(HERE), "include", f) for f in ("hx_sim.h", "hx_ppo.h")]
OUT = os.path.join(HERE, "libhx.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SRC:
        obj = src[:-4] + ".o"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-array-bounds",
               "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
