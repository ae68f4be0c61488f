#!/usr/bin/env python3
"""One learner GEMM shape, repeated (for rocprofv3 --pmc runs).  args: kind bk rows out in_ld iters"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd import capi
L = capi.lib()
kind, bk, rows, out, in_ld, iters = (int(x) for x in sys.argv[1:7])
ms = C.c_float()
capi.check(L.hx_ppo_gemm_bench(kind, bk, rows, out, in_ld, iters, C.byref(ms)))
print(f"kind={kind} bk={bk} rows={rows} out={out} in={in_ld}: {ms.value*1e3:.1f} us {2.0*rows*out*in_ld/ms.value/1e9:.1f} TF")
