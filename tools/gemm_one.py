#!/usr/bin/env python3
"""One learner GEMM shape, few launches (for rocprofv3 --pmc runs).  usage: gemm_one.py kind bk out in [rows] [iters]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd import capi
L = capi.lib()
kind, bk, o, i = (int(x) for x in sys.argv[1:5])
rows = int(sys.argv[5]) if len(sys.argv) > 5 else 61440
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ms = C.c_float()
capi.check(L.hx_ppo_gemm_bench(kind, bk, rows, o, i, iters, C.byref(ms)))
print(f"kind={kind} bk={bk} out={o} in={i} rows={rows}: {ms.value*1e3:.1f} us {2.0*rows*o*i/ms.value/1e9:.1f} TF")
