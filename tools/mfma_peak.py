#!/usr/bin/env python3
"""Where the fp32 matrix-pipe throughput goes: probe kernels that add the GEMM's ingredients one at a time
(include/hx_ppo.h hx_mfma_probe).  GPU box only.  usage: mfma_peak.py [mfmas_per_wave]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd import capi
L = capi.lib()
L.hx_mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
names = ["MFMA only", "+ LDS fragment reads", "+ barrier per BK16 tile", "+ global loads and LDS stores", "3 with [k][rows] b32 fragment reads", "3 with guarded global loads"]
for mode in range(6):
    for wgs_per_cu in (1, 2, 3, 4):
        tf = C.c_float()
        capi.check(L.hx_mfma_probe(mode, 256 * wgs_per_cu, n, C.byref(tf)), "probe")
        print(f"mode {mode} {names[mode]:32s} waves per SIMD {wgs_per_cu}: {tf.value:6.1f} TFLOP/s  ({tf.value / 157.3:.2f} of peak)")
