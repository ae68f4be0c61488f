#!/usr/bin/env python3
"""Per-layer timing of the learner's GEMMs (hx_ppo_gemm_bench) for both K-depths.  GPU box only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_amd import capi
L = capi.lib()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
layers = [(512, 616), (256, 512), (128, 256), (768, 1052), (256, 768), (128, 256)]
for bk in (16, 32, 116, 132):
    tot = {0: [0, 0], 1: [0, 0], 2: [0, 0]}
    for kind, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
        if (rows < 16384 and kind != 0) or (bk > 100 and kind == 2):
            continue
        for li, (o, i) in enumerate(layers):
            if kind == 1 and li % 3 == 0:
                continue                      # no dX for the first layer
            ms = C.c_float()
            capi.check(L.hx_ppo_gemm_bench(kind, bk, rows, o, i, 20, C.byref(ms)))
            fl = 2.0 * rows * o * i
            tot[kind][0] += fl; tot[kind][1] += ms.value
            print(f"bk={bk} {name:6s} rows={rows} out={o:4d} in={i:5d} {ms.value*1e3:8.1f} us {fl/ms.value/1e9:7.1f} TF")
    for kind, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad")):
        if tot[kind][1]:
            print(f"== bk={bk} {name} total {tot[kind][1]:.3f} ms {tot[kind][0]/tot[kind][1]/1e9:.1f} TF")
