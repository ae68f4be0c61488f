"""Register / scratch budget of the env-step kernel, read from the device assembly hipcc emits for gfx950 (no GPU needed).

The kernel runs one wave per SIMD and is latency bound; a kernel that touches scratch memory at all also pays ~9 us per launch on
this part (tools/micro/launch_gap.hip).  Round 4 brought all three robot models to zero scratch (DESIGN.md 3.2,
profiles/r04_bd_env_registers.txt): what had held the registers was the compiler keeping every state field's address from its load to
its store.  This test keeps it that way."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_env_step_kernels_use_no_scratch(tmp_path):
    from isaac_amd import build
    src = os.path.join(ROOT, "isaac_amd", "csrc", "hx_sim.hip")
    out = str(tmp_path / "hx_sim.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-array-bounds", *build.EXTRA_FLAGS["hx_sim.hip"],
           '-DHX_BUILD_ID="t"', "--cuda-device-only", "-S", src, "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = open(out).read()
    # amdhsa.kernels metadata: one block per kernel with .name, .private_segment_fixed_size, .vgpr_count, .vgpr_spill_count
    blocks = re.findall(r"\.name:\s+(\S+)\s(?:.*\n)*?.*?\.private_segment_fixed_size:\s+(\d+)(?:.*\n)*?.*?\.vgpr_count:\s+(\d+)(?:.*\n)*?.*?\.vgpr_spill_count:\s+(\d+)", text)
    env = {name: (int(scratch), int(vgpr), int(spill)) for name, scratch, vgpr, spill in blocks if "hx_env_step_kernel" in name}
    assert len(env) == 3, sorted(env)
    for name, (scratch, vgpr, spill) in env.items():
        print(name, "scratch", scratch, "registers", vgpr, "spilled", spill)
        assert scratch == 0 and spill == 0, f"{name}: {scratch} B of scratch, {spill} spilled registers"
        assert vgpr <= 448, f"{name}: {vgpr} registers (hector 312, humanoid_ppo 341, hector_full 408 in round 4)"
