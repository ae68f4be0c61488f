"""GPU: the f32 MFMA GEMM template (isaac_amd/csrc/hx_gemm.h) against numpy float64, all three modes,
including ragged M/N/K that exercise every bounds guard.  Tolerance: fp32 dot products of length K,
|err| <= 2e-6 * sum|a*b| (MFMA f32 is an exact fma chain; numpy reference is float64)."""
import numpy as np
import pytest

from isaac_amd import capi

pytestmark = pytest.mark.gpu


def _run(hxlib, mode, M, N, K, rng):
    r4 = lambda x: (x + 3) // 4 * 4
    if mode in (0, 3, 8, 9):      # FWD: Y = elu(X W^T + b); X [M][K], W [N][K]  (8 / 9: on the persistent grid)
        lda, ldb, ldc = r4(K), r4(K), N
        A = np.zeros((M, lda), np.float32); A[:, :K] = rng.standard_normal((M, K))
        B = np.zeros((N, ldb), np.float32); B[:, :K] = rng.standard_normal((N, K)) / np.sqrt(K)
        bias = rng.standard_normal(N).astype(np.float32)
        z = A[:, :K].astype(np.float64) @ B[:, :K].astype(np.float64).T + bias
        ref = np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
        H = None
        Kk = lda
    elif mode in (1, 4):    # DGRAD: dX = (dZ W) * elu'(H); dZ [M][K], W [K][N]
        lda, ldb, ldc = K, N, N
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
        H = rng.standard_normal((M, N)).astype(np.float32)
        H = np.where(H > 0, H, np.expm1(np.minimum(H, 0))).astype(np.float32)
        bias = None
        ref = (A.astype(np.float64) @ B.astype(np.float64)) * np.where(H > 0, 1.0, H.astype(np.float64) + 1.0)
        Kk = K
    else:                   # WGRAD: dW[M][N] = dZ[K][M]^T X[K][N]
        lda, ldb, ldc = M, N, N
        A = rng.standard_normal((K, M)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        H = None
        bias = np.zeros(M, np.float32)
        ref = A.astype(np.float64).T @ B.astype(np.float64)
        Kk = K
    dA, dB = capi.DeviceBuffer.from_host(A), capi.DeviceBuffer.from_host(B)
    dC = capi.DeviceBuffer(M * ldc * 4)
    dbias = capi.DeviceBuffer.from_host(bias) if bias is not None else None
    dH = capi.DeviceBuffer.from_host(H) if H is not None else None
    capi.check(hxlib.hx_ppo_gemm_test(mode, M, N, Kk, dA.ptr, lda, dB.ptr, ldb, dbias.ptr if dbias else None,
                                      dC.ptr, ldc, dH.ptr if dH else None, None), "gemm_test")
    out = dC.download(np.float32, (M, ldc))[:, :N]
    scale = np.abs(ref).max() + 1.0
    err = np.abs(out - ref).max()
    assert err <= 5e-5 * scale * max(1.0, np.sqrt(K) / 16), (mode, M, N, K, err, scale)
    if mode == 2:
        db = dbias.download(np.float32, (M,))
        np.testing.assert_allclose(db, A.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3 * np.sqrt(K))


@pytest.mark.parametrize("mode,M,N,K", [
    (0, 256, 256, 64), (0, 4096, 512, 615), (0, 1000, 768, 1050), (0, 61, 130, 20),
    (3, 64, 128, 615), (3, 4096, 256, 512), (3, 70, 100, 36),
    (1, 512, 256, 128), (1, 1000, 512, 256), (4, 4096, 768, 256), (4, 33, 132, 8),
    (2, 512, 616, 2048), (2, 768, 1052, 1000), (2, 128, 256, 960), (2, 100, 36, 77),
])
def test_gemm_modes(hxlib, mode, M, N, K):
    _run(hxlib, mode, M, N, K, np.random.default_rng(mode * 1000 + M + N + K))


@pytest.mark.parametrize("mode,M,N,K", [(8, 8192, 768, 1052), (8, 1000, 256, 768), (8, 70, 100, 36), (9, 8192, 256, 256), (9, 333, 130, 615)])
def test_persistent_forward(hxlib, mode, M, N, K):
    """The background critic's persistent-grid forward product (hx_gemm_persistent_kernel; 7 workgroups walking all tiles)."""
    _run(hxlib, mode, M, N, K, np.random.default_rng(mode * 1000 + M + N + K))


def test_mfma_layout_asymmetric(hxlib):
    """A = I with an asymmetric B catches a transposed C/D register map (guide: 'A=I-check with ASYMMETRIC B')."""
    M = N = K = 128
    A = np.eye(M, dtype=np.float32)
    W = (np.arange(N)[:, None] * 1000 + np.arange(K)[None, :]).astype(np.float32) / 1.0e5    # W[n][k]
    bias = np.zeros(N, np.float32)
    dA, dB, db = capi.DeviceBuffer.from_host(A), capi.DeviceBuffer.from_host(W), capi.DeviceBuffer.from_host(bias)
    dC = capi.DeviceBuffer(M * N * 4)
    capi.check(hxlib.hx_ppo_gemm_test(0, M, N, K, dA.ptr, K, dB.ptr, K, db.ptr, dC.ptr, N, None, None), "gemm")
    out = dC.download(np.float32, (M, N))
    np.testing.assert_allclose(out, W.T, rtol=0, atol=1e-6)      # elu is the identity for positive inputs


@pytest.mark.parametrize("mode,M,N,K", [(10, 1000, 768, 1050), (13, 4096, 256, 512), (11, 1000, 512, 256), (14, 33, 132, 8),
                                        (12, 768, 1052, 1000), (12, 100, 36, 77)])
def test_gemm_modes_bk32(hxlib, mode, M, N, K):
    """The BK = 32 instantiations (mode + 10) against the same float64 references."""
    import types
    real = hxlib.hx_ppo_gemm_test
    shim = types.SimpleNamespace(hx_ppo_gemm_test=lambda m, *a: real(m + 10, *a))
    _run(shim, mode - 10, M, N, K, np.random.default_rng(mode + M + N + K))


# ---------------------------------------------------------------------------------------------- bf16-input kernels
def _bf16_round(x):
    """fp32 -> bf16 round-to-nearest-even -> fp32 (what v_cvt_pk_bf16_f32 does), in numpy."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


@pytest.mark.parametrize("mode,M,N,K", [(5, 256, 256, 64), (5, 4096, 512, 616), (5, 1000, 768, 1052), (5, 61, 130, 20),
                                        (6, 512, 256, 128), (6, 1000, 512, 256), (6, 33, 132, 8)])
def test_bf16_gemm_modes(hxlib, mode, M, N, K):
    """hx_gemm_bf16.h: operands rounded to bf16 (RNE) on load, fp32 accumulation.  Against float64 numpy on the
    SAME rounded operands the only difference is fp32 summation order: same bound as the fp32 kernels."""
    rng = np.random.default_rng(M + N + K)
    if mode == 5:           # forward: Y = elu(X W^T + b), X [M][K], W [N][K]
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        z = _bf16_round(A).astype(np.float64) @ _bf16_round(B).astype(np.float64).T + bias
        ref = np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
        H = None
    else:                   # dgrad: dX = (dZ W) * elu'(H), dZ [M][K], W^T given K-major as [N][K]
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        H = rng.standard_normal((M, N)).astype(np.float32)
        H = np.where(H > 0, H, np.expm1(np.minimum(H, 0))).astype(np.float32)
        bias = None
        ref = (_bf16_round(A).astype(np.float64) @ _bf16_round(B).astype(np.float64).T) * np.where(H > 0, 1.0, H.astype(np.float64) + 1.0)
    dA, dB = capi.DeviceBuffer.from_host(A), capi.DeviceBuffer.from_host(B)
    dC = capi.DeviceBuffer(M * N * 4)
    dbias = capi.DeviceBuffer.from_host(bias) if bias is not None else None
    dH = capi.DeviceBuffer.from_host(H) if H is not None else None
    capi.check(hxlib.hx_ppo_gemm_test(mode, M, N, K, dA.ptr, K, dB.ptr, K, dbias.ptr if dbias else None, dC.ptr, N,
                                      dH.ptr if dH else None, None), "gemm_test")
    out = dC.download(np.float32, (M, N))
    scale = np.abs(ref).max() + 1.0
    err = np.abs(out - ref).max()
    assert err <= 5e-5 * scale * max(1.0, np.sqrt(K) / 16), (mode, M, N, K, err, scale)
    # and the rounding itself is the bf16 one: against the unrounded product the error is ~2^-9 relative per operand
    if mode == 5:
        z32 = A.astype(np.float64) @ B.astype(np.float64).T + bias
        full = np.where(z32 > 0, z32, np.expm1(np.minimum(z32, 0)))
        rel = np.abs(out - full).max() / scale
        assert 1e-5 < rel < 2e-2, rel


@pytest.mark.parametrize("M,N,K", [(512, 616, 2048), (768, 1052, 1000), (128, 256, 960), (100, 36, 77)])
def test_bf16_wgrad(hxlib, M, N, K):
    """hx_wgrad_bf16_kernel: dW[M][N] = round_bf16(dZ[K][M])^T round_bf16(X[K][N]) with fp32 accumulation (k-pair packed
    LDS image); the bias gradient (column sums of dZ) comes from the unrounded fp32 values."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    ref = _bf16_round(A).astype(np.float64).T @ _bf16_round(B).astype(np.float64)
    dA, dB = capi.DeviceBuffer.from_host(A), capi.DeviceBuffer.from_host(B)
    dC = capi.DeviceBuffer(M * N * 4)
    dbias = capi.DeviceBuffer.from_host(np.zeros(M, np.float32))
    capi.check(hxlib.hx_ppo_gemm_test(7, M, N, K, dA.ptr, M, dB.ptr, N, dbias.ptr, dC.ptr, N, None, None), "gemm_test")
    out = dC.download(np.float32, (M, N))
    scale = np.abs(ref).max() + 1.0
    assert np.abs(out - ref).max() <= 5e-5 * scale * max(1.0, np.sqrt(K) / 16)
    np.testing.assert_allclose(dbias.download(np.float32, (M,)), A.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3 * np.sqrt(K))


@pytest.mark.parametrize("layers,rows,slots", [
    # the hector networks' six products at a quarter of the update's rows: 768 x 1052 = 768 x 1024 in 256 x 256 tiles + a 28-column strip
    ([(768, 1052), (512, 616), (256, 768), (256, 512), (128, 256), (128, 256)], 15360, 0),
    # humanoid_ppo's input layers (708 / 220 wide), few slots: several slices per workgroup round and edge tiles in both directions
    ([(512, 708), (768, 220), (256, 512)], 4096, 64),
    # ragged everything: out not a multiple of 128, widths that leave 4 and 36 columns over
    ([(100, 260), (36, 36), (132, 1060)], 1024, 40),
])
def test_wgrad_one_workgroup_per_cu(hxlib, layers, rows, slots):
    """hx_wgrad_multi_kernel + hx_wgrad_plan.h + the 2-D slab reduction against numpy float64: dW_l = dZ_l^T X_l and the bias
    gradient (column sums of dZ_l), every layer of a launch group in its own tile shape and slice count.  The slabs are poisoned
    with NaN before the launches, so an element no workgroup writes fails the comparison."""
    import ctypes as C
    rng = np.random.default_rng(rows + len(layers))
    nl = len(layers)
    dZ = [rng.standard_normal((rows, o)).astype(np.float32) for o, _ in layers]
    X = [rng.standard_normal((rows, i)).astype(np.float32) for _, i in layers]
    bz = [capi.DeviceBuffer.from_host(a) for a in dZ]
    bx = [capi.DeviceBuffer.from_host(a) for a in X]
    bw = [capi.DeviceBuffer(o * i * 4) for o, i in layers]
    bb = [capi.DeviceBuffer(o * 4) for o, _ in layers]
    arr = lambda bufs: (C.c_void_p * nl)(*[b.ptr for b in bufs])
    ints = lambda v: (C.c_int * nl)(*v)
    nlaunch = C.c_int(0)
    capi.check(hxlib.hx_ppo_wgrad_multi_test(nl, ints([o for o, _ in layers]), ints([i for _, i in layers]), rows, arr(bz), arr(bx), arr(bw), arr(bb),
                                             slots, C.byref(nlaunch), None), "wgrad_multi_test")
    assert 1 <= nlaunch.value <= 2
    for l, (o, i) in enumerate(layers):
        ref = dZ[l].astype(np.float64).T @ X[l].astype(np.float64)
        out = bw[l].download(np.float32, (o, i))
        assert np.isfinite(out).all(), l
        err = np.abs(out - ref).max()
        assert err <= 5e-5 * (np.abs(ref).max() + 1.0) * max(1.0, np.sqrt(rows) / 16), (l, err)
        np.testing.assert_allclose(bb[l].download(np.float32, (o,)), dZ[l].astype(np.float64).sum(0), rtol=1e-4, atol=1e-3 * np.sqrt(rows))
