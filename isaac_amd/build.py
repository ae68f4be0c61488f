"""Build libhx.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("hx_sim.hip", "hx_ppo.hip", "hx_comm.hip")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("hx_math.h", "hx_dyn.h", "hx_env.h", "hx_gemm.h", "hx_gemm_bf16.h", "hx_common.h", "hx_model_data.h", "hx_model_data_full.h", "hx_model_data_xbot.h")] + \
      [os.path.join(os.path.dirname(HERE), "include", f) for f in ("hx_sim.h", "hx_ppo.h")]
OUT = os.path.join(HERE, "libhx.so")
# per-file flags chosen by measurement (profiles/): see DESIGN.md "Env-step kernel"
EXTRA_FLAGS = {
    # env-step kernel: SLP-packing fp32 into v_pk_* costs more register moves and spills than it saves in this
    # latency-bound, register-heavy kernel (288 -> 194 us per step), and relaxed fp32 (rcp/rsq instead of IEEE
    # division sequences, contraction) brings it to 162 us; parity tests run with exactly these flags.
    "hx_sim.hip": ["-fno-slp-vectorize", "-ffast-math"],
}


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + HDR)


def build(force=False, verbose=False):
    """Serialised across processes with a file lock: under `torch.distributed.run` every rank calls this (bench.py), and
    ranks that found the library stale at the same time must not compile into the same files concurrently."""
    import fcntl
    with open(OUT + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SRC:
        obj = src[:-4] + ".o"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-array-bounds"]
        cmd += EXTRA_FLAGS.get(os.path.basename(src), [])
        cmd += os.environ.get("HX_EXTRA_FLAGS_" + os.path.basename(src).split(".")[0].upper(), "").split()
        cmd += ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
