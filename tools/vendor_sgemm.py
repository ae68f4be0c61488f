#!/usr/bin/env python3
"""Context for the learner's GEMM numbers: what does the vendor library (rocBLAS / hipBLASLt through torch.mm, fp32, TF32 off)
reach on the same three products of every hidden layer at the update's size (61 440 rows)?  GPU box only.
Prints TFLOP/s per layer and product beside the fp32-MFMA peak (157.3); the learner's own kernels: bench.py roofline.all_gemm_kernels."""
import time
import torch

assert torch.cuda.is_available()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
M = 61440
layers = [("actor 616->512", 616, 512), ("actor 512->256", 512, 256), ("actor 256->128", 256, 128),
          ("critic 1052->768", 1052, 768), ("critic 768->256", 768, 256), ("critic 256->128", 256, 128)]


def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


print("%-18s %12s %12s %12s   (TFLOP/s, fp32, peak 157.3)" % ("layer", "fwd X W^T", "dgrad dZ W", "wgrad dZ^T X"))
tot_f = tot_t = 0.0
for name, K, N in layers:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dZ = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    tf = bench(lambda: torch.mm(X, W.t()))
    td = bench(lambda: torch.mm(dZ, W))
    tw = bench(lambda: torch.mm(dZ.t(), X))
    tot_f += 3 * fl; tot_t += tf + td + tw
    print("%-18s %12.1f %12.1f %12.1f" % (name, fl / tf / 1e12, fl / td / 1e12, fl / tw / 1e12))
print("all 18 products: %.1f TFLOP/s = %.3f of the fp32-MFMA peak" % (tot_f / tot_t / 1e12, tot_f / tot_t / 1e12 / 157.3))
