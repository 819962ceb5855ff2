"""Kernel-level parity (-m gpu): every libpdmk entry point, called through the C ABI, against a plain fp32 torch
statement of the same op on identical inputs.  Tolerances: fp32 path 2e-4 of the output scale (MFMA fp32 is an exact
fmaf chain; only summation order differs); bf16 path 2e-2 of the output scale against fp32 math on the bf16-rounded
inputs (bf16 storage of outputs/intermediates, fp32 accumulation)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f32": 2e-4, "bf16": 2e-2}


def close(got, ref, tol, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert math.isfinite(err) and err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"
    # max-norm alone is lenient where most of the output is small next to its largest element: bound the relative L2 error too
    l2 = (got - ref).double().norm().item() / (ref.double().norm().item() + 1e-30)
    assert l2 <= tol, f"{what}: relative L2 error {l2:.3e} (tol {tol})"


def rnd(shape, dev, dtype, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def conv_w_pack(w):   # [Co,Ci,3,3] -> [Co, 9*Ci]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(200, 136, 96), (128, 128, 32), (1, 8, 32), (515, 320, 1024)])
def test_gemm_linear(dev, dn, M, N, K):
    from pdm import _pdmk as k
    torch.manual_seed(0)
    dt = DT[dn]
    A, B = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5)
    bias = torch.randn(N, device=dev)
    nb = 1 if M < 5 else 5
    rows_per_b = (M + nb - 1) // nb
    rv = torch.randn(nb, N, device=dev)
    R = rnd((M, N), dev, dt)
    C = torch.zeros(M, N, device=dev, dtype=dt)
    k.gemm(A, B, C, M, N, K, K, K, N, bias=bias, rowvec=rv, rows_per_b=rows_per_b, R=R, ldr=N)
    ref = A.float() @ B.float().t() + bias + rv[torch.arange(M, device=dev) // rows_per_b] + R.float()
    close(C, ref, TOL[dn], "gemm+epilogue")
    C2 = C.clone()
    k.gemm(A, B, C2, M, N, K, K, K, N, accumulate=True, alpha=0.5)
    close(C2, C.float() + 0.5 * (A.float() @ B.float().t()), TOL[dn], "gemm accumulate")
    Cf = torch.zeros(M, N, device=dev)
    k.gemm(A, B, Cf, M, N, K, K, K, N, out_f32=True)
    close(Cf, A.float() @ B.float().t(), TOL[dn], "gemm out_f32")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_gemm_conv_fwd_modes(dev, dn, mode):
    from pdm import _pdmk as k
    torch.manual_seed(1)
    dt = DT[dn]
    Bn, Ci, Co, Hs = 2, 32, 40, 10
    x = rnd((Bn, Hs, Hs, Ci), dev, dt)          # NHWC source
    w = rnd((Co, Ci, 3, 3), dev, dt, (9 * Ci) ** -0.5)
    xn = x.float().permute(0, 3, 1, 2)
    if mode == 0:
        ref = F.conv2d(xn, w.float(), padding=1)
    elif mode == 1:
        ref = F.conv2d(xn, w.float(), stride=2, padding=1)
    elif mode == 2:
        ref = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode="nearest"), w.float(), padding=1)
    else:   # transposed stride 2 with the same (un-flipped) taps: y[v] = sum_t x[(v+t-1)/2] w[t]
        up = torch.zeros(Bn, Ci, 2 * Hs, 2 * Hs, device=dev)
        up[:, :, ::2, ::2] = xn
        ref = F.conv2d(up, w.float(), padding=1)
    Ho = ref.shape[2]
    M = Bn * Ho * Ho
    C = torch.zeros(M, Co, device=dev, dtype=dt)
    k.gemm(x, conv_w_pack(w), C, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV,
           conv=(Bn, Hs, Hs, Ci, Ho, Ho, mode, Ci))
    close(C.view(Bn, Ho, Ho, Co), ref.permute(0, 2, 3, 1), TOL[dn], f"conv mode {mode}")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("splitk", [1, 3])
def test_gemm_wgrad_linear(dev, dn, splitk):
    from pdm import _pdmk as k
    torch.manual_seed(2)
    dt = DT[dn]
    P, No, Ki = 300, 96, 160            # P pixels (reduction), dW [No, Ki]
    dY, X = rnd((P, No), dev, dt), rnd((P, Ki), dev, dt)
    dW = torch.zeros(No, Ki, device=dev)
    db = torch.ones(No, device=dev)
    k.gemm(dY, X, dW, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=splitk,
           colsum_out=db)
    close(dW, dY.float().t() @ X.float(), TOL[dn], "wgrad linear")
    close(db, 1.0 + dY.float().sum(0), TOL[dn], "fused bias gradient")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gemm_wgrad_conv(dev, dn, mode):
    from pdm import _pdmk as k
    torch.manual_seed(3)
    dt = DT[dn]
    Bn, Ci, Co, Hs = 2, 32, 64, 8
    x = rnd((Bn, Hs, Hs, Ci), dev, dt)
    xn = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(Co, Ci, 3, 3, device=dev, requires_grad=True)
    if mode == 0:
        y = F.conv2d(xn, w, padding=1)
    elif mode == 1:
        y = F.conv2d(xn, w, stride=2, padding=1)
    else:
        y = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode="nearest"), w, padding=1)
    Ho = y.shape[2]
    dy = rnd((Bn, Ho, Ho, Co), dev, dt)
    (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
    P = Bn * Ho * Ho
    dW = torch.zeros(Co, 9 * Ci, device=dev)
    db = torch.zeros(Co, device=dev)
    k.gemm(dy, x, dW, Co, 9 * Ci, P, Co, 0, 9 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=2,
           conv=(Bn, Hs, Hs, Ci, Ho, Ho, mode, Ci), colsum_out=db)
    close(dW, conv_w_pack(gw), TOL[dn], f"wgrad conv mode {mode}")
    close(db, dy.float().sum(dim=(0, 1, 2)), TOL[dn], "fused conv bias gradient")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_conv_dgrad_via_cast_permute(dev, dn):
    from pdm import _pdmk as k
    torch.manual_seed(4)
    dt = DT[dn]
    Bn, Ci, Co, Hs = 2, 32, 64, 8
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
    wp = conv_w_pack(w)                                   # fp32 master layout [Co, 9, Ci]
    wd = torch.zeros(Ci, 9 * Co, device=dev, dtype=dt)
    k.cast_permute(wp, wd, Co, 9, Ci, 2)
    wq = wp.to(dt).float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2)    # the rounded weights, OIHW
    for stride, mode in ((1, 0), (2, 3)):
        xn = torch.randn(Bn, Ci, Hs, Hs, device=dev, requires_grad=True)
        y = F.conv2d(xn, wq, stride=stride, padding=1)
        Hy = y.shape[2]
        dy = rnd((Bn, Hy, Hy, Co), dev, dt)
        (gx,) = torch.autograd.grad(y, xn, dy.float().permute(0, 3, 1, 2))
        M = Bn * Hs * Hs
        dX = torch.zeros(M, Ci, device=dev, dtype=dt)
        k.gemm(dy, wd, dX, M, Ci, 9 * Co, 0, 9 * Co, Ci, a_mode=k.A_CONV, conv=(Bn, Hy, Hy, Co, Hs, Hs, mode, Co))
        close(dX.view(Bn, Hs, Hs, Ci), gx.permute(0, 2, 3, 1), TOL[dn], f"conv dgrad stride {stride}")
    # Linear transpose copy
    W = torch.randn(40, 24, device=dev)
    Wt = torch.zeros(24, 40, device=dev, dtype=dt)
    k.cast_permute(W, Wt, 40, 1, 24, 1)
    close(Wt, W.t().to(dt), 0, "cast_permute transpose")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("C,G,gs,HW,silu", [(64, 32, 2, 256, True), (40, 17, 2, 64, True), (320, 32, 10, 1024, False),
                                            (2560, 32, 80, 64, True), (176, 17, 10, 4, True), (320, 32, 10, 4096, True),
                                            (3072, 32, 96, 16, False)])
def test_groupnorm(dev, dn, C, G, gs, HW, silu):
    from pdm import _pdmk as k
    torch.manual_seed(5)
    dt = DT[dn]
    Bn, cr = 3, G * gs
    x = rnd((Bn, HW, C), dev, dt) + 0.5
    gamma, beta = torch.randn(cr, device=dev) * 0.3 + 1, torch.randn(cr, device=dev) * 0.3
    y = torch.full((Bn, HW, C), 7.0, device=dev, dtype=dt)
    stats = torch.zeros(Bn, G, 2, device=dev)
    ws = torch.zeros(Bn * G * 64, device=dev, dtype=torch.float64)
    k.groupnorm_fwd(x, y, gamma, beta, stats, ws, Bn, HW, C, C, C, G, gs, 1e-5, silu)
    xr = x.float()[..., :cr].permute(0, 2, 1).clone().requires_grad_(True)      # [B, cr, HW]
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.group_norm(xr, G, gr, br, 1e-5)
    ref = F.silu(z) if silu else z
    close(y[..., :cr], ref.permute(0, 2, 1), TOL[dn], "gn fwd")
    assert (y[..., cr:] == 0).all()
    dy = rnd((Bn, HW, C), dev, dt)
    gx, gg, gb = torch.autograd.grad(ref, [xr, gr, br], dy.float()[..., :cr].permute(0, 2, 1))
    dx = torch.zeros_like(x)
    dgm, dbt = torch.zeros(cr, device=dev), torch.zeros(cr, device=dev)
    k.groupnorm_bwd(x, dy, dx, gamma, beta, stats, dgm, dbt, ws, Bn, HW, C, C, C, C, G, gs, silu, False)
    close(dx[..., :cr], gx.permute(0, 2, 1), TOL[dn] * 2, "gn dx")
    close(dgm, gg, TOL[dn] * 2, "gn dgamma")
    close(dbt, gb, TOL[dn] * 2, "gn dbeta")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("C,M", [(64, 37), (320, 300), (1280, 64)])
def test_layernorm(dev, dn, C, M):
    from pdm import _pdmk as k
    torch.manual_seed(6)
    dt = DT[dn]
    x = rnd((M, C), dev, dt) + 0.3
    gamma, beta = torch.randn(C, device=dev) * 0.3 + 1, torch.randn(C, device=dev) * 0.3
    y = torch.zeros_like(x)
    stats = torch.zeros(M, 2, device=dev)
    k.layernorm_fwd(x, y, gamma, beta, stats, M, C, C, C, 1e-5)
    xr, gr, br = x.float().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    close(y, ref, TOL[dn], "ln fwd")
    dy = rnd((M, C), dev, dt)
    gx, gg, gb = torch.autograd.grad(ref, [xr, gr, br], dy.float())
    dx = torch.zeros_like(x)
    dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    k.layernorm_bwd(x, dy, dx, gamma, stats, dgm, dbt, M, C, C, C, C, False)
    close(dx, gx, TOL[dn] * 2, "ln dx")
    close(dgm, gg, TOL[dn] * 2, "ln dgamma")
    close(dbt, gb, TOL[dn] * 2, "ln dbeta")


@pytest.mark.parametrize("cand", [-1, 0, 1, 2, 3, 5, 6, 7])
def test_wgrad_slabs_and_grouped_finish(dev, force_cfg, cand):
    """Weight gradients whose splits store partial slabs (accumulate = 2 on the reduction-major path) and
    pdmk_splitk_finish_group adding the slabs of many weights in one launch: equal to fp32 math, and bit-reproducible
    (two runs give identical gradients - the atomic form does not promise that); 35 queued weights flush in two launches;
    the planner's slab-mode split factor is used; error cases of the C entry point."""
    import ctypes as C
    from pdm import _pdmk as k
    if cand >= 0:
        force_cfg("PDMK_WGRAD_CFG", cand)
    torch.manual_seed(33)
    dt = torch.bfloat16
    runs = []
    for rep in range(2):
        q = k.SlabQueue()
        outs = []
        torch.manual_seed(34)
        for i in range(35):
            P, No, Ki = ((4096, 320, 352), (2048, 96, 160), (8192, 64, 64), (1000, 160, 288))[i % 4]
            dY, X = rnd((P, No), dev, dt), rnd((P, Ki), dev, dt)
            dW = torch.full((No, Ki), 1.0, device=dev)
            db = torch.ones(No, device=dev)
            k.wgrad(dY, X, dW, No, Ki, P, No, Ki, colsum_out=db, queue=q)
            outs.append((dY, X, dW, db))
        assert len(q.items) <= 35
        q.flush()
        torch.cuda.synchronize()
        runs.append(outs)
    for (dY, X, dW, db), (_, _, dW2, _) in zip(*runs):
        close(dW, 1.0 + dY.float().t() @ X.float(), 2e-2, "slab wgrad")
        close(db, 1.0 + dY.float().sum(0), 2e-2, "bias gradient beside slabs")
        assert torch.equal(dW, dW2), "slab sums are added in a fixed order"
    # 3x3 conv weight gradients through the same queue (the halo kernels, candidates 6 / 7, store slabs too)
    q = k.SlabQueue()
    for Bn, Hs, Ci, Co in ((2, 32, 64, 96), (4, 16, 96, 160), (1, 64, 32, 64)):
        x = rnd((Bn, Hs, Hs, Ci), dev, dt)
        w = torch.zeros(Co, Ci, 3, 3, device=dev, requires_grad=True)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1)
        dy = rnd((Bn, Hs, Hs, Co), dev, dt)
        (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
        P = Bn * Hs * Hs
        dW = torch.full((Co, 9 * Ci), 1.0, device=dev)
        db = torch.ones(Co, device=dev)
        k.wgrad(dy, x, dW, Co, 9 * Ci, P, Co, 0, b_mode=k.B_COLK_CONV, conv=(Bn, Hs, Hs, Ci, Hs, Hs, 0, Ci), colsum_out=db, queue=q)
        q.flush()
        close(dW, 1.0 + conv_w_pack(gw), 2e-2, f"slab conv wgrad B{Bn} {Hs}x{Hs} {Ci}->{Co}")
        close(db, 1.0 + dy.float().sum(dim=(0, 1, 2)), 2e-2, "conv bias gradient beside slabs")
    item = (k.SlabItem * 1)()
    assert k._lib.pdmk_splitk_finish_group(C.cast(item, C.c_void_p), 1, None) == -1
    assert k._lib.pdmk_splitk_finish_group(C.cast(item, C.c_void_p), 33, None) == -1


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_deferred_partial_reductions(dev, dn):
    """PartialQueue / pdmk_reduce_partials_group: GroupNorm and LayerNorm backward passes that leave their affine-gradient
    partials in slabs of their own, reduced by ONE launch per <= 32 layers, give the gradients of the immediate two-stage form
    (same slabs, same adds per address; fp32 atomics of <= 8 slices: order-dependent in the last bits only) - 40 queued layers
    of mixed shapes (more than one group), accumulation into non-zero gradients, and the error cases of the C entry point."""
    import ctypes as C
    from pdm import _pdmk as k
    torch.manual_seed(41)
    dt = DT[dn]
    q = k.PartialQueue()
    cases = []
    for i in range(40):
        if i % 2 == 0:
            Cc, G, gs, HW, Bn = ((320, 32, 10, 256, 2), (64, 32, 2, 64, 3), (176, 17, 10, 16, 1))[(i // 2) % 3]
            x = rnd((Bn, HW, Cc), dev, dt) + 0.5
            gamma, beta = torch.randn(G * gs, device=dev) * 0.3 + 1, torch.randn(G * gs, device=dev) * 0.3
            y, stats = torch.zeros_like(x), torch.zeros(Bn, G, 2, device=dev)
            ws = torch.zeros(Bn * G * 64, device=dev, dtype=torch.float64)
            k.groupnorm_fwd(x, y, gamma, beta, stats, ws, Bn, HW, Cc, Cc, Cc, G, gs, 1e-5, True)
            dy = rnd((Bn, HW, Cc), dev, dt)
            outs = []
            for queue in (None, q):
                dx = torch.zeros_like(x)
                dg, db = torch.full((G * gs,), 0.5, device=dev), torch.full((G * gs,), -0.25, device=dev)
                k.groupnorm_bwd(x, dy, dx, gamma, beta, stats, dg, db, ws, Bn, HW, Cc, Cc, Cc, Cc, G, gs, True, False, queue=queue)
                outs.append((dx, dg, db))
        else:
            M, Cc = ((300, 320), (64, 1280), (37, 64))[(i // 2) % 3]
            x = rnd((M, Cc), dev, dt) + 0.3
            gamma, beta = torch.randn(Cc, device=dev) * 0.3 + 1, torch.randn(Cc, device=dev) * 0.3
            y, stats = torch.zeros_like(x), torch.zeros(M, 2, device=dev)
            k.layernorm_fwd(x, y, gamma, beta, stats, M, Cc, Cc, Cc, 1e-5)
            dy = rnd((M, Cc), dev, dt)
            outs = []
            for queue in (None, q):
                dx = torch.zeros_like(x)
                dg, db = torch.full((Cc,), 0.5, device=dev), torch.full((Cc,), -0.25, device=dev)
                k.layernorm_bwd(x, dy, dx, gamma, stats, dg, db, M, Cc, Cc, Cc, Cc, False, queue=queue)
                outs.append((dx, dg, db))
        cases.append(outs)
    assert 0 < len(q.items) < 40          # the queue flushed itself once when it reached 32 items
    q.flush()
    assert not q.items
    torch.cuda.synchronize()
    for (dx0, dg0, db0), (dx1, dg1, db1) in cases:
        assert torch.equal(dx0, dx1)
        for a, b in ((dg0, dg1), (db0, db1)):
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max() + 1e-6), "deferred reduction differs"
    item = (k.PartialItem * 1)()
    assert k._lib.pdmk_reduce_partials_group(C.cast(item, C.c_void_p), 1, None) == -1        # null slab
    assert k._lib.pdmk_reduce_partials_group(C.cast(item, C.c_void_p), 33, None) == -1       # more than PDMK_PARTIAL_GROUP_MAX


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("Nq,Nk,H", [(100, 100, 3), (64, 77, 2), (4, 4, 1), (256, 13, 5), (1024, 77, 2), (4096, 13, 1),
                                     (4096, 4096, 2), (9216, 9216, 1), (9216, 77, 2)])   # 9216 = 96x96 latents (BASELINE configs[4])
def test_attention(dev, dn, Nq, Nk, H):
    from pdm import _pdmk as k
    torch.manual_seed(7)
    dt = DT[dn]
    Bn, D = 2, 64
    self_attn = Nq == Nk
    # fused projections output layout: q|k|v side by side in one row when self-attention
    if self_attn:
        qkv = rnd((Bn, Nq, 3 * H * D), dev, dt)
        q, kk, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
        qs = ks = vs = (Nq * 3 * H * D, 3 * H * D)
    else:
        q = rnd((Bn, Nq, H * D), dev, dt)
        kv = rnd((Bn, Nk, 2 * H * D), dev, dt)
        kk, v = kv[..., :H * D], kv[..., H * D:]
        qs, ks, vs = (Nq * H * D, H * D), (Nk * 2 * H * D, 2 * H * D), (Nk * 2 * H * D, 2 * H * D)
    o = torch.zeros(Bn, Nq, H * D, device=dev, dtype=dt)
    lse = torch.zeros(Bn, H, Nq, device=dev)
    os_ = (Nq * H * D, H * D)
    scale = D ** -0.5
    k.attn_fwd(q, kk, v, o, lse, Bn, H, Nq, Nk, qs, ks, vs, os_, scale)
    qr = q.float().reshape(Bn, Nq, H, D).transpose(1, 2).clone().requires_grad_(True)
    kr = kk.float().reshape(Bn, Nk, H, D).transpose(1, 2).clone().requires_grad_(True)
    vr = v.float().reshape(Bn, Nk, H, D).transpose(1, 2).clone().requires_grad_(True)
    s = (qr @ kr.transpose(-1, -2)) * scale
    ref = torch.softmax(s, -1) @ vr
    close(o.view(Bn, Nq, H, D).transpose(1, 2), ref, TOL[dn], "attn fwd")
    close(lse * math.log(2.0), torch.logsumexp(s, -1), TOL[dn], "attn lse")
    do = rnd((Bn, Nq, H * D), dev, dt)
    gq, gk, gv = torch.autograd.grad(ref, [qr, kr, vr], do.float().view(Bn, Nq, H, D).transpose(1, 2))
    dq = torch.zeros(Bn, Nq, H * D, device=dev, dtype=dt)
    dk = torch.zeros(Bn, Nk, H * D, device=dev, dtype=dt)
    dv = torch.zeros(Bn, Nk, H * D, device=dev, dtype=dt)
    delta = torch.zeros(Bn, H, Nq, device=dev)
    k.attn_bwd(q, kk, v, o, do, lse, delta, dq, dk, dv, Bn, H, Nq, Nk, qs, ks, vs, os_, os_, (Nk * H * D, H * D),
               (Nk * H * D, H * D), scale)
    close(dq.view(Bn, Nq, H, D).transpose(1, 2), gq, TOL[dn] * 2, "attn dq")
    close(dk.view(Bn, Nk, H, D).transpose(1, 2), gk, TOL[dn] * 2, "attn dk")
    close(dv.view(Bn, Nk, H, D).transpose(1, 2), gv, TOL[dn] * 2, "attn dv")


@pytest.mark.parametrize("Nq,Nk", [(128, 77), (200, 200), (256, 13)])
def test_attention_strongly_negative_scores_with_a_key_tail(dev, Nq, Nk):
    """Rows whose real scores are ALL far below zero (log-sum-exp < -128 in base 2) next to key blocks with missing keys: a
    missing key has K = 0, i.e. score 0 and exp2(0 - lse) = inf - the masked tail block of the forward and of the dQ kernel must
    keep that out of the sums (an unmasked inf * 0 poisons the whole dQ row; found by the step-parity suite after a large-lr
    update), and the wide / narrow forms of all three kernels must agree with fp32 math."""
    from pdm import _pdmk as k
    torch.manual_seed(19)
    dt = torch.bfloat16
    Bn, H, D = 2, 2, 64
    u = torch.ones(D, device=dev)
    q = (4.5 * u + 0.25 * torch.randn(Bn, Nq, H, D, device=dev)).to(dt).reshape(Bn, Nq, H * D)
    kk = (-4.5 * u + 0.25 * torch.randn(Bn, Nk, H, D, device=dev)).to(dt).reshape(Bn, Nk, H * D)
    v = rnd((Bn, Nk, H * D), dev, dt)
    qs, ks, os_ = (Nq * H * D, H * D), (Nk * H * D, H * D), (Nq * H * D, H * D)
    scale = D ** -0.5
    o = torch.zeros(Bn, Nq, H * D, device=dev, dtype=dt)
    lse = torch.zeros(Bn, H, Nq, device=dev)
    k.attn_fwd(q, kk, v, o, lse, Bn, H, Nq, Nk, qs, ks, ks, os_, scale)
    assert float(lse.max()) < -128.0, float(lse.max())          # the regime under test
    qr = q.float().reshape(Bn, Nq, H, D).transpose(1, 2).clone().requires_grad_(True)
    kr = kk.float().reshape(Bn, Nk, H, D).transpose(1, 2).clone().requires_grad_(True)
    vr = v.float().reshape(Bn, Nk, H, D).transpose(1, 2).clone().requires_grad_(True)
    sc = (qr @ kr.transpose(-1, -2)) * scale
    ref = torch.softmax(sc, -1) @ vr
    close(o.view(Bn, Nq, H, D).transpose(1, 2), ref, 2e-2, "attn fwd, negative scores")
    do = rnd((Bn, Nq, H * D), dev, dt)
    gq, gk, gv = torch.autograd.grad(ref, [qr, kr, vr], do.float().view(Bn, Nq, H, D).transpose(1, 2))
    dq, dk, dv = (torch.zeros(Bn, n, H * D, device=dev, dtype=dt) for n in (Nq, Nk, Nk))
    delta = torch.zeros(Bn, H, Nq, device=dev)
    k.attn_bwd(q, kk, v, o, do, lse, delta, dq, dk, dv, Bn, H, Nq, Nk, qs, ks, ks, os_, os_, ks, ks, scale)
    for name, got in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert torch.isfinite(got.float()).all(), name
    close(dq.view(Bn, Nq, H, D).transpose(1, 2), gq, 8e-2, "attn dq, negative scores")   # (ill-conditioned on purpose)
    close(dk.view(Bn, Nk, H, D).transpose(1, 2), gk, 8e-2, "attn dk, negative scores")
    close(dv.view(Bn, Nk, H, D).transpose(1, 2), gv, 4e-2, "attn dv, negative scores")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_elementwise_family(dev, dn):
    from pdm import _pdmk as k
    torch.manual_seed(8)
    dt = DT[dn]
    tol = TOL[dn]
    # GEGLU
    M, Fd = 50, 120
    x = rnd((M, 2 * Fd), dev, dt)
    y = torch.zeros(M, Fd, device=dev, dtype=dt)
    k.geglu_fwd(x, y, M, Fd, 2 * Fd, Fd)
    xr = x.float().requires_grad_(True)
    ref = xr[:, :Fd] * F.gelu(xr[:, Fd:])
    close(y, ref, tol, "geglu fwd")
    dy = rnd((M, Fd), dev, dt)
    (gx,) = torch.autograd.grad(ref, xr, dy.float())
    dx = torch.zeros_like(x)
    k.geglu_bwd(x, dy, dx, M, Fd, 2 * Fd, Fd, 2 * Fd)
    close(dx, gx, tol, "geglu bwd")
    # SiLU
    t = rnd((3, 1280), dev, dt)
    st = torch.zeros_like(t)
    k.silu_fwd(t, st)
    tr = t.float().requires_grad_(True)
    close(st, F.silu(tr), tol, "silu")
    dyt = rnd((3, 1280), dev, dt)
    dxt = torch.zeros_like(t)
    k.silu_bwd(t, dyt, dxt)
    close(dxt, torch.autograd.grad(F.silu(tr), tr, dyt.float())[0], tol, "silu bwd")
    # copy2d into / out of a concat buffer, accumulate
    a, b = rnd((20, 24), dev, dt), rnd((20, 40), dev, dt)
    cat = torch.zeros(20, 64, device=dev, dtype=dt)
    k.copy2d(a, cat, 20, 24, 24, 64)
    k.copy2d(b, cat[:, 24:], 20, 40, 40, 64)
    assert torch.equal(cat, torch.cat([a, b], 1))
    k.copy2d(b, cat[:, 24:], 20, 40, 40, 64, accumulate=True)
    close(cat[:, 24:], 2 * b.float(), tol, "copy2d acc")
    # colsum
    xs = rnd((333, 48), dev, dt)
    cs = torch.zeros(48, device=dev)
    k.colsum(xs, cs, 333, 48, 48)
    close(cs, xs.float().sum(0), tol, "colsum")
    # pool
    src = rnd((2, 8, 8, 16), dev, dt)
    dst = torch.zeros(2, 4, 4, 16, device=dev, dtype=dt)
    k.pool2x2_sum(src, dst, 2, 4, 4, 16)
    close(dst, F.avg_pool2d(src.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1) * 4, tol, "pool")
    # axpby
    xa, ya = rnd((1000,), dev, dt), rnd((1000,), dev, dt)
    r = 2.0 * xa.float() - 1.0 * ya.float()
    k.axpby(xa, ya, 2.0, -1.0)
    close(ya, r, tol, "axpby")


def test_scalar_kernels(dev):
    from pdm import _pdmk as k
    import numpy as np, os
    torch.manual_seed(9)
    # timestep embedding against the reference's CompVis twin (golden fixture)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_twins.npz"))
    t = torch.from_numpy(gold["temb_t"]).to(dev)
    out = torch.zeros(len(t), 320, device=dev)
    freqs = torch.exp(-math.log(10000.0) * torch.arange(160, dtype=torch.float32) / 160).to(dev)
    k.timestep_embed(t, freqs, out, len(t), 320)
    # tolerance: sin/cos of a fp32 argument up to 999 rad is conditioned at |x| * 2^-24 = 6e-5 (one ulp of the argument)
    close(out, torch.from_numpy(gold["temb_ref"]).to(dev), 1e-4, "timestep_embed vs ldm twin")
    # forward diffusion
    Bn, Cc, HW = 3, 4, 64
    x0, nz = torch.randn(Bn, Cc, HW, device=dev), torch.randn(Bn, Cc, HW, device=dev)
    ts = torch.tensor([0, 500, 999], device=dev)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
    ac = torch.cumprod(1 - betas, 0).to(dev)
    sa, sb = ac.sqrt().contiguous(), (1 - ac).sqrt().contiguous()
    noisy = torch.zeros(Bn, HW, 8, device=dev)
    target = torch.zeros(Bn, HW, 8, device=dev)
    k.add_noise_velocity(x0, nz, ts, sa, sb, noisy, target, Bn, Cc, HW, 8)
    a, s = sa[ts].view(-1, 1, 1), sb[ts].view(-1, 1, 1)
    close(noisy[..., :4], (a * x0 + s * nz).permute(0, 2, 1), 1e-6, "add_noise")
    close(target[..., :4], (a * nz - s * x0).permute(0, 2, 1), 1e-6, "velocity")
    assert (noisy[..., 4:] == 0).all()
    back = torch.zeros(Bn, Cc, HW, device=dev)
    k.nhwc_to_nchw(noisy, back, Bn, Cc, HW, 8)
    close(back, a * x0 + s * nz, 1e-6, "nhwc_to_nchw")
    nh = torch.zeros(Bn, HW, 8, device=dev, dtype=torch.bfloat16)
    k.nchw_to_nhwc(x0, nh, Bn, Cc, HW, 8)
    close(nh[..., :4], x0.permute(0, 2, 1).bfloat16(), 0, "nchw_to_nhwc")
    # mse fwd/bwd with per-sample weights, mixed dtypes
    pa = torch.randn(Bn, 100, 8, device=dev).bfloat16()
    pb = torch.randn(Bn, 100, 8, device=dev)
    w = torch.rand(Bn, device=dev)
    out = torch.zeros(4, device=dev, dtype=torch.float64)
    k.mse_fwd(pa, pb, w, out, 2, Bn, 100, 4, 8, 8, 1.0 / (Bn * 400))
    ref = (((pa.float()[..., :4] - pb[..., :4]) ** 2).mean(dim=(1, 2)) * w).mean()
    close(out[2:3], ref.view(1), 1e-5, "mse fwd")
    da = torch.zeros_like(pa)
    k.mse_bwd(pa, pb, w, da, Bn, 100, 4, 8, 8, 8, 0.37, False)
    close(da[..., :4], 0.37 * w.view(-1, 1, 1) * (pa.float()[..., :4] - pb[..., :4]), 1e-2, "mse bwd")
    # AdamW against torch.optim.AdamW over 3 steps
    n = 1024 * 4 + 8
    p0 = torch.randn(n, device=dev)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr = torch.tensor([1e-3], device=dev)
    for stp in range(1, 4):
        g = torch.randn(n, device=dev)
        pt.grad = g.clone()
        opt.step()
        bc = torch.tensor([1 - 0.9 ** stp, 1 - 0.999 ** stp], device=dev)
        gg = g.clone()
        wb = torch.zeros(n, device=dev, dtype=torch.bfloat16)
        k.adamw(p, gg, m, v, n, lr, 0.9, 0.999, 1e-8, 0.01, bc, 1.0, True, w_bf16=wb)
        assert torch.equal(wb, p.bfloat16())
        assert (gg == 0).all()
    close(p, pt.detach(), 1e-6, "adamw")
    ss = torch.zeros(2, device=dev, dtype=torch.float64)
    k.sumsq(p0, n, ss, 1)
    close(ss[1:2], (p0.double() ** 2).sum().view(1), 1e-6, "sumsq")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(8, 1280, 320), (2, 136, 64), (8, 336, 1280), (16, 64, 96), (1, 8, 32)])
def test_skinny_gemm_and_wgrad(dev, dn, M, N, K):
    """pdmk_skinny_gemm / pdmk_skinny_wgrad (time-embedding linears, M = batch) vs fp32 torch."""
    from pdm import _pdmk as k
    torch.manual_seed(5)
    dt = DT[dn]
    x, w = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5)
    bias = torch.randn(N, device=dev)
    ref = x.float() @ w.float().t() + bias
    for ydt in (torch.float32, dt):
        y = torch.full((M, N), 7.0, device=dev, dtype=ydt)
        k.skinny_gemm(x, w, y, M, N, K, K, K, N, bias=bias)
        close(y, ref, TOL[dn], "skinny gemm")
        y2 = y.clone()
        k.skinny_gemm(x, w, y2, M, N, K, K, K, N, accumulate=True)
        close(y2, y.float() + x.float() @ w.float().t(), TOL[dn], "skinny gemm accumulate")
    # dgrad form: fp32 skinny operand against compute-dtype weights
    dyf = torch.randn(M, N, device=dev)
    wt = w.t().contiguous()
    dx = torch.zeros(M, K, device=dev, dtype=dt)
    k.skinny_gemm(dyf, wt, dx, M, K, N, N, N, K)
    close(dx, dyf @ w.float(), TOL[dn], "skinny dgrad")
    for dy in (dyf, dyf.to(dt)) if dn == "bf16" else (dyf,):
        dw = torch.ones(N, K, device=dev)
        db = torch.ones(N, device=dev)
        k.skinny_wgrad(dy, x, dw, db, M, N, K, N, K, K)
        close(dw, 1.0 + dy.float().t() @ x.float(), TOL[dn], "skinny wgrad")
        close(db, 1.0 + dy.float().sum(0), TOL[dn], "skinny bias grad")


@pytest.fixture
def force_cfg():
    """Force a GEMM candidate (0 = K-step-32 kernels, 1.. = LDS-DMA ring shapes) instead of the tuned plan."""
    import os
    def setter(var, val):
        os.environ[var] = str(val)
    yield setter
    for var in ("PDMK_RING_CFG", "PDMK_WGRAD_CFG"):
        os.environ.pop(var, None)


@pytest.mark.parametrize("cand", list(range(13)) + [17, 18, 19, 20, 21])
def test_ring_gemm_every_tile_shape(dev, force_cfg, cand):
    """Every fwd/dgrad candidate of the plan cache gives the same linear + 3x3-conv results (ragged M/N/K tails, K not
    a multiple of the K-step, stride-2 / upsample / transposed gathers, split-K with fp32 atomics)."""
    from pdm import _pdmk as k
    force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(7)
    dt = torch.bfloat16
    for M, N, K in ((515, 352, 608), (200, 136, 96), (64, 32, 32), (1000, 1280, 160)):
        A, B = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5)
        bias, R = torch.randn(N, device=dev), rnd((M, N), dev, dt)
        C = torch.full((M, N), 3.0, device=dev, dtype=dt)
        k.gemm(A, B, C, M, N, K, K, K, N, bias=bias, R=R, ldr=N)
        ref = A.float() @ B.float().t() + bias + R.float()
        close(C, ref, 2e-2, f"linear {M}x{N}x{K}")
        Cf = torch.zeros(M, N, device=dev)
        k.gemm(A, B, Cf, M, N, K, K, K, N, out_f32=True, splitk=3)
        close(Cf, A.float() @ B.float().t(), 2e-2, f"linear split-K {M}x{N}x{K}")
    Bn, Ci, Co, Hs = 2, 96, 72, 12
    x = rnd((Bn, Hs, Hs, Ci), dev, dt)
    w = rnd((Co, Ci, 3, 3), dev, dt, (9 * Ci) ** -0.5)
    xn = x.float().permute(0, 3, 1, 2)
    for mode in (0, 1, 2):
        if mode == 0:
            ref = F.conv2d(xn, w.float(), padding=1)
        elif mode == 1:
            ref = F.conv2d(xn, w.float(), stride=2, padding=1)
        else:
            ref = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode="nearest"), w.float(), padding=1)
        Ho = ref.shape[2]
        y = torch.zeros(Bn * Ho * Ho, Co, device=dev, dtype=dt)
        k.gemm(x, conv_w_pack(w), y, Bn * Ho * Ho, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV,
               conv=(Bn, Hs, Hs, Ci, Ho, Ho, mode, Ci))
        close(y, ref.permute(0, 2, 3, 1).reshape(-1, Co), 2e-2, f"conv mode {mode}")


@pytest.mark.parametrize("cand", [6, 7])
def test_halo_conv_wgrad(dev, force_cfg, cand):
    """conv_wgrad_halo_kernel (dY tile + input patch staged once per 128-pixel block, nine taps out of the patch) vs
    autograd: row-group blocks (64^2, 32^2, 16^2, odd batches), the 8^2 fall-back, channel / output-row tails, one split (overwrite and accumulate) and several (atomics)."""
    from pdm import _pdmk as k
    force_cfg("PDMK_WGRAD_CFG", cand)
    torch.manual_seed(12)
    dt = torch.bfloat16
    for Bn, Hs, Ci, Co, sk, acc in ((2, 64, 96, 160, 4, True), (3, 16, 64, 96, 1, False), (1, 32, 160, 64, 1, True),
                                    (2, 16, 32, 320, 2, True), (5, 16, 352, 96, 3, True), (1, 64, 32, 32, 7, True),
                                    (3, 8, 64, 96, 2, True)):
        x = rnd((Bn, Hs, Hs, Ci), dev, dt)
        w = torch.zeros(Co, Ci, 3, 3, device=dev, requires_grad=True)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1)
        dy = rnd((Bn, Hs, Hs, Co), dev, dt)
        (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
        P = Bn * Hs * Hs
        base = 1.0 if (acc or sk > 1) else 0.0
        dW = torch.full((Co, 9 * Ci), 1.0, device=dev)
        db = torch.ones(Co, device=dev)
        k.gemm(dy, x, dW, Co, 9 * Ci, P, Co, 0, 9 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=sk,
               accumulate=acc, conv=(Bn, Hs, Hs, Ci, Hs, Hs, 0, Ci), colsum_out=db)
        used = k.candidate_name(k.A_COLK, k.B_COLK_CONV, k.last_candidate()).startswith("pdmk_ring::conv_wgrad_halo_kernel")
        assert used == (Hs >= 16), f"{Hs}x{Hs}: halo wgrad eligibility"      # 8x8 images (two per block) exceed the patch buffer
        close(dW, base + conv_w_pack(gw), 2e-2, f"halo wgrad B{Bn} {Hs}x{Hs} {Ci}->{Co} sk{sk}")
        close(db, 1.0 + dy.float().sum(dim=(0, 1, 2)), 2e-2, "fused conv bias gradient")


@pytest.mark.parametrize("cand", [0, 1, 2, 3, 4, 5, 6, 7])
def test_ring_wgrad_candidates(dev, force_cfg, cand):
    from pdm import _pdmk as k
    force_cfg("PDMK_WGRAD_CFG", cand)
    torch.manual_seed(8)
    dt = torch.bfloat16
    for P, No, Ki, sk in ((300, 96, 160, 1), (4096, 320, 352, 5), (1000, 160, 288, 2)):
        dY, X = rnd((P, No), dev, dt), rnd((P, Ki), dev, dt)
        dW = torch.ones(No, Ki, device=dev)
        db = torch.ones(No, device=dev)
        k.gemm(dY, X, dW, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=sk,
               accumulate=(sk == 1), colsum_out=db)
        close(dW, 1.0 + dY.float().t() @ X.float(), 2e-2, f"wgrad linear P{P}")
        close(db, 1.0 + dY.float().sum(0), 2e-2, "fused bias gradient")
    Bn, Ci, Co, Hs = 2, 96, 160, 12
    x = rnd((Bn, Hs, Hs, Ci), dev, dt)
    xn = x.float().permute(0, 3, 1, 2)
    for mode in (0, 1, 2):
        w = torch.zeros(Co, Ci, 3, 3, device=dev, requires_grad=True)
        if mode == 0:
            y = F.conv2d(xn, w, padding=1)
        elif mode == 1:
            y = F.conv2d(xn, w, stride=2, padding=1)
        else:
            y = F.conv2d(F.interpolate(xn, scale_factor=2.0, mode="nearest"), w, padding=1)
        Ho = y.shape[2]
        dy = rnd((Bn, Ho, Ho, Co), dev, dt)
        (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
        P = Bn * Ho * Ho
        dW = torch.zeros(Co, 9 * Ci, device=dev)
        db = torch.zeros(Co, device=dev)
        k.gemm(dy, x, dW, Co, 9 * Ci, P, Co, 0, 9 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=2,
               conv=(Bn, Hs, Hs, Ci, Ho, Ho, mode, Ci), colsum_out=db)
        close(dW, conv_w_pack(gw), 2e-2, f"wgrad conv mode {mode}")
        close(db, dy.float().sum(dim=(0, 1, 2)), 2e-2, "fused conv bias gradient")


@pytest.mark.parametrize("cand", [13, 14, 15, 16])
def test_halo_conv_2d_tiles_on_wide_images(dev, force_cfg, cand):
    """Images wider than a tile (the VAE encoder's 128^2 .. 512^2 levels): the halo kernel cuts them into rows x tw blocks;
    output rows of a tile are then not contiguous (2-D row map in the epilogue).  Bias + residual, and split-K slabs."""
    from pdm import _pdmk as k
    force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(10)
    dt = torch.bfloat16
    for Bn, Hs, Ws, Ci, Co, sk in ((1, 128, 128, 64, 40, 1), (2, 32, 256, 32, 160, 1), (1, 16, 512, 96, 72, 1),
                                   (1, 64, 128, 160, 64, 2)):
        x = rnd((Bn, Hs, Ws, Ci), dev, dt)
        w = rnd((Co, Ci, 3, 3), dev, dt, (9 * Ci) ** -0.5)
        bias = torch.randn(Co, device=dev)
        M = Bn * Hs * Ws
        R = rnd((M, Co), dev, dt)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        if sk == 1:
            y = torch.zeros(M, Co, device=dev, dtype=dt)
            k.gemm(x, conv_w_pack(w), y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(Bn, Hs, Ws, Ci, Hs, Ws, 0, Ci),
                   bias=bias, R=R, ldr=Co)
            ref = ref + bias + R.float()
        else:
            ws = torch.full((sk, M, Co), 7.0, device=dev)
            k.gemm(x, conv_w_pack(w), ws, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(Bn, Hs, Ws, Ci, Hs, Ws, 0, Ci),
                   out_f32=True, splitk=sk, accumulate=2)
            y = ws.sum(0)
        assert k.candidate_name(k.A_CONV, k.B_ROWK, k.last_candidate()).startswith("pdmk_ring::conv_halo_kernel"), \
            f"{Hs}x{Ws}: halo candidate {cand} was not eligible"
        close(y, ref, 2e-2, f"halo conv 2-D tiles B{Bn} {Hs}x{Ws} {Ci}->{Co} sk{sk}")


@pytest.mark.parametrize("cand", [13, 14, 15, 16])
def test_halo_conv_candidates(dev, force_cfg, cand):
    """conv_halo_kernel (input patch staged once per 64-channel block, 9 taps out of LDS) vs F.conv2d: row-group tiles,
    whole-image tiles, several images per tile with a missing last image, channel tails, split-K over channel blocks."""
    from pdm import _pdmk as k
    force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(9)
    dt = torch.bfloat16
    for Bn, Hs, Ci, Co, sk in ((2, 16, 96, 72, 1), (3, 8, 64, 160, 1), (1, 32, 160, 64, 1), (2, 64, 32, 320, 1), (2, 16, 160, 136, 2)):
        x = rnd((Bn, Hs, Hs, Ci), dev, dt)
        w = rnd((Co, Ci, 3, 3), dev, dt, (9 * Ci) ** -0.5)
        bias = torch.randn(Co, device=dev)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias if sk == 1 else None, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        M = Bn * Hs * Hs
        if sk == 1:
            y = torch.zeros(M, Co, device=dev, dtype=dt)
            k.gemm(x, conv_w_pack(w), y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(Bn, Hs, Hs, Ci, Hs, Hs, 0, Ci), bias=bias)
        else:
            y = torch.zeros(M, Co, device=dev)
            k.gemm(x, conv_w_pack(w), y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(Bn, Hs, Hs, Ci, Hs, Hs, 0, Ci),
                   out_f32=True, splitk=sk)
        close(y, ref, 2e-2, f"halo conv B{Bn} {Hs}x{Hs} {Ci}->{Co} sk{sk}")


@pytest.mark.parametrize("cand", [20, 21])
@pytest.mark.parametrize("grp", [0, 1, 3])
def test_rowblock_linear(dev, force_cfg, cand, grp):
    """rowblock_kernel (A rows x all of K as register fragments, weight tiles streamed through one ring over the n-tiles of a
    column group, wave-private epilogue with the residual prefetched one n-tile ahead) vs fp32 math and - bit for bit - vs the
    ring kernel: ragged M / N / K tails, strided A / C / R rows, bias, residual, accumulate, one and several column groups,
    more n-tiles than ring slots, a single K-step."""
    import os
    from pdm import _pdmk as k
    torch.manual_seed(31)
    dt = torch.bfloat16
    kmax = 320 if cand == 20 else 640
    if grp:
        os.environ["PDMK_RB_GRP"] = str(grp)
    try:
        for M, N, K, lda, ldc, ldr, mode in ((1000, 960, 320, 320, 960, 960, "bias+res"), (515, 352, 608, 616, 360, 352, "res"),
                                             (4096, 2560, 320, 328, 2560, 0, "bias"), (300, 136, 96, 96, 136, 144, "acc"),
                                             (129, 1288, 640, 640, 1288, 0, "plain"), (64, 8, 32, 32, 8, 8, "bias+res"),
                                             (2048, 320, 64, 64, 328, 320, "bias+acc")):
            if K > kmax:
                continue
            A = rnd((M, lda), dev, dt)
            B = rnd((N, K), dev, dt, K ** -0.5)
            bias = torch.randn(N, device=dev) if "bias" in mode else None
            R = rnd((M, ldr), dev, dt) if "res" in mode else None
            outs = []
            for c in (cand, 4):
                force_cfg("PDMK_RING_CFG", c)
                C = torch.full((M, ldc), 3.0, device=dev, dtype=dt)
                k.gemm(A, B, C, M, N, K, lda, K, ldc, bias=bias, R=R, ldr=ldr if R is not None else 0, accumulate="acc" in mode)
                name = k.candidate_name(k.A_ROWK, k.B_ROWK, k.last_candidate())
                assert name.startswith("pdmk_rb::rowblock_kernel" if c == cand else "pdmk_ring::igemm_ring_kernel"), name
                outs.append(C)
            ref = A[:, :K].float() @ B.float().t()
            if bias is not None:
                ref = ref + bias
            if R is not None:
                ref = ref + R[:, :N].float()
            if "acc" in mode:
                ref = ref + 3.0
            close(outs[0][:, :N], ref, 2e-2, f"rowblock {M}x{N}x{K} {mode}")
            assert torch.equal(outs[0], outs[1]), (M, N, K, mode, (outs[0].float() - outs[1].float()).abs().max().item())
            if ldc > N:
                assert float((outs[0][:, N:].float() - 3.0).abs().max()) == 0.0      # columns past N untouched
    finally:
        os.environ.pop("PDMK_RB_GRP", None)


@pytest.mark.parametrize("grp", [0, 1, 3])
def test_layernorm_in_the_gemm_prologue(dev, grp):
    """pdmk_gemm_args.ln_gamma: LayerNorm(A) @ W^T (+ bias, + the GEGLU epilogue) as ONE launch of the row-block kernel - the row
    block is normalised in registers - against pdmk_layernorm_fwd followed by the same GEMM (forced row-block candidate: the
    normalised operand differs only where the two kernels' fp32 row sums round differently, so the products agree to bf16
    resolution) and against fp32 torch; the optional outputs of the training path (mean / rstd and the normalised rows) against
    pdmk_layernorm_fwd's; ragged M, K below the register image (zero-padded chunks must stay zero), several column groups, a
    strided A.  BasicTransformerBlock norm1 -> to_q/k/v, norm2 -> to_q, norm3 -> ff.net.0.proj (blocks.py:705-867)."""
    import os
    from pdm import _pdmk as k
    torch.manual_seed(77)
    dt = torch.bfloat16
    if grp:
        os.environ["PDMK_RB_GRP"] = str(grp)
    try:
        for M, N, K, lda, mode in ((300, 320, 320, 320, "plain"), (4096, 960, 320, 328, "plain"), (1024, 640, 640, 640, "bias"),
                                   (515, 352, 256, 264, "bias"), (2048, 512, 320, 320, "geglu"), (1000, 1408, 320, 320, "geglu+keep"),
                                   (256, 1920, 640, 640, "plain"), (64, 8, 32, 32, "bias")):
            A = (rnd((M, lda), dev, dt) * 1.7 + 0.4)
            W = rnd((N, K), dev, dt, K ** -0.5)
            gamma, beta = torch.randn(K, device=dev) * 0.3 + 1, torch.randn(K, device=dev) * 0.3
            bias = torch.randn(N, device=dev) if ("bias" in mode or "geglu" in mode) else None
            geglu = "geglu" in mode
            assert k.gemm_ln_supported(A, W, M, N, K, lda, K, geglu=geglu, bias=bias is not None), (M, N, K, mode)
            # two launches: LayerNorm, then the GEMM through the same kernel family
            ln_ref = torch.zeros(M, K, device=dev, dtype=dt)
            st_ref = torch.zeros(M, 2, device=dev)
            k.layernorm_fwd(A, ln_ref, gamma, beta, st_ref, M, K, lda, K, 1e-5)
            st, lno = torch.zeros(M, 2, device=dev), torch.full((M, K + 8), 5.0, device=dev, dtype=dt)
            ln = (gamma, beta, st, lno, 1e-5)
            os.environ["PDMK_RING_CFG"] = "20" if K <= 320 else "21"
            try:
                if geglu:
                    gl0, gl1 = torch.zeros(M, N // 2, device=dev, dtype=dt), torch.zeros(M, N // 2, device=dev, dtype=dt)
                    pre0 = torch.zeros(M, N, device=dev, dtype=dt) if "keep" in mode else None
                    pre1 = torch.zeros(M, N, device=dev, dtype=dt) if "keep" in mode else None
                    assert k.gemm_geglu(ln_ref, W, gl0, pre0, M, N, K, K, K, bias=bias)
                    assert k.gemm_geglu(A, W, gl1, pre1, M, N, K, lda, K, bias=bias, ln=ln)
                    y0, y1 = gl0, gl1
                    z = ln_ref.float() @ W.float().t() + bias
                    zz = z.reshape(M, N // 16, 2, 8).to(dt).float()
                    ref = (zz[:, :, 0] * F.gelu(zz[:, :, 1])).reshape(M, N // 2)
                    if pre0 is not None:
                        close(pre1, pre0, 1e-2, "pre-activation copy")
                else:
                    y0, y1 = torch.zeros(M, N, device=dev, dtype=dt), torch.full((M, N + 8), 3.0, device=dev, dtype=dt)
                    k.gemm(ln_ref, W, y0, M, N, K, K, K, N, bias=bias)
                    k.gemm(A, W, y1, M, N, K, lda, K, N + 8, bias=bias, ln=ln)
                    assert k.candidate_name(k.A_ROWK, k.B_ROWK, k.last_candidate()).startswith("pdmk_rb::rowblock_kernel")
                    assert float((y1[:, N:].float() - 3.0).abs().max()) == 0.0
                    y1 = y1[:, :N]
                    ref = ln_ref.float() @ W.float().t() + (bias if bias is not None else 0.0)
            finally:
                os.environ.pop("PDMK_RING_CFG", None)
            close(y1, ref, 2e-2, f"LN-prologue GEMM vs fp32 {M}x{N}x{K} {mode}")
            close(y1, y0, 6e-3, f"LN-prologue GEMM vs LayerNorm + GEMM {M}x{N}x{K} {mode}")
            close(st, st_ref, 1e-5, "mean / rstd from the prologue")
            close(lno[:, :K], ln_ref, 8e-3, "normalised rows from the prologue")         # <= one bf16 ulp where the sums round apart
            mism = (lno[:, :K] != ln_ref).float().mean().item()
            assert mism < 0.02, mism                                                      # ... and only on a few elements
            assert float((lno[:, K:].float() - 5.0).abs().max()) == 0.0                  # columns past K untouched
            # inference form: no statistics, no normalised rows
            y2 = torch.zeros_like(y1)
            if geglu:
                assert k.gemm_geglu(A, W, y2, None, M, N, K, lda, K, bias=bias, ln=(gamma, beta, None, None, 1e-5))
            else:
                k.gemm(A, W, y2, M, N, K, lda, K, N, bias=bias, ln=(gamma, beta, None, None, 1e-5))
            assert torch.equal(y2, y1.contiguous())
        # shapes the row-block kernel does not take are refused (-2), never computed without the LayerNorm
        A, W = rnd((256, 1280), dev, dt), rnd((320, 1280), dev, dt)
        assert not k.gemm_ln_supported(A, W, 256, 320, 1280, 1280, 1280)
        with pytest.raises(k.PdmkError):
            k.gemm(A, W, torch.zeros(256, 320, device=dev, dtype=dt), 256, 320, 1280, 1280, 1280, 320,
                   ln=(torch.ones(1280, device=dev), torch.zeros(1280, device=dev), None, None, 1e-5))
    finally:
        os.environ.pop("PDMK_RB_GRP", None)


def _interleave8(h, g):
    """[.., F] hidden and gate -> [.., 2F] with (hidden, gate) interleaved in blocks of 8 columns (PDMK_EPI_GEGLU layout)."""
    F_ = h.shape[-1]
    return torch.stack([h.reshape(*h.shape[:-1], F_ // 8, 8), g.reshape(*g.shape[:-1], F_ // 8, 8)], dim=-2).reshape(*h.shape[:-1], 2 * F_)


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_geglu_interleaved_layout(dev, dn):
    """layout 1 of pdmk_geglu_fwd / _bwd: hidden / gate interleaved in blocks of 8 columns, strided rows."""
    from pdm import _pdmk as k
    torch.manual_seed(21)
    dt = DT[dn]
    M, Fd, ld = 37, 48, 104
    h, g = rnd((M, Fd), dev, dt), rnd((M, Fd), dev, dt)
    x = torch.zeros(M, ld, device=dev, dtype=dt)
    x[:, :2 * Fd] = _interleave8(h, g)
    y = torch.zeros(M, Fd, device=dev, dtype=dt)
    k.geglu_fwd(x, y, M, Fd, ld, Fd, layout=1)
    hr, gr = h.float().requires_grad_(True), g.float().requires_grad_(True)
    ref = hr * F.gelu(gr)
    close(y, ref, TOL[dn], "geglu layout 1")
    dy = rnd((M, Fd), dev, dt)
    gh, gg = torch.autograd.grad(ref, [hr, gr], dy.float())
    dx = torch.zeros(M, ld, device=dev, dtype=dt)
    k.geglu_bwd(x, dy, dx, M, Fd, ld, Fd, ld, layout=1)
    close(dx[:, :2 * Fd], _interleave8(gh, gg), TOL[dn] * 2, "geglu bwd layout 1")
    assert float(dx[:, 2 * Fd:].abs().max()) == 0.0


@pytest.mark.parametrize("cand", [1, 4, 6, 8, 9, 10, 12, 17, 18, 19, 20, 21, -1])
def test_gemm_fused_geglu_epilogue(dev, force_cfg, cand):
    """PDMK_EPI_GEGLU: hidden * gelu(gate) formed in the projection's epilogue equals - bit for bit - the projection stored
    in bf16 followed by pdmk_geglu_fwd(layout 1), for every ring tile shape (ragged M, N tails of 16, K tail), with and
    without the pre-activation copy; the K-step-32 kernels have no such epilogue (status -2 -> False)."""
    from pdm import _pdmk as k
    if cand >= 0:
        force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(22)
    dt = torch.bfloat16
    for M, N, K in ((515, 352, 608), (200, 48, 96), (64, 32, 32), (1000, 1296, 160)):
        if cand == 20 and K > 320:       # row-block candidate 20 keeps K <= 320 in registers (21: K <= 640)
            continue
        A, B = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5)
        bias = torch.randn(N, device=dev)
        two = torch.zeros(M, N, device=dev, dtype=dt)
        k.gemm(A, B, two, M, N, K, K, K, N, bias=bias)
        ref = torch.zeros(M, N // 2, device=dev, dtype=dt)
        k.geglu_fwd(two, ref, M, N // 2, N, N // 2, layout=1)
        for keep in (True, False):
            gl = torch.full((M, N // 2 + 8), 5.0, device=dev, dtype=dt)[:, :N // 2]      # strided output rows
            f = torch.full((M, N), 7.0, device=dev, dtype=dt) if keep else None
            assert k.gemm_geglu(A, B, gl, f, M, N, K, K, K, bias=bias)
            assert torch.equal(gl, ref), (M, N, K, keep, (gl.float() - ref.float()).abs().max().item())
            if keep:
                assert torch.equal(f, two)
        # against fp32 math on the same inputs
        pre = (A.float() @ B.float().t() + bias).to(dt).float()
        hg = pre.reshape(M, N // 16, 2, 8)
        close(ref, (hg[:, :, 0] * F.gelu(hg[:, :, 1])).reshape(M, N // 2), 2e-2, "fused geglu vs fp32")
    os_environ = __import__("os").environ
    os_environ["PDMK_RING_CFG"] = "0"
    gl = torch.zeros(64, 16, device=dev, dtype=dt)
    assert k.gemm_geglu(rnd((64, 32), dev, dt), rnd((32, 32), dev, dt), gl, None, 64, 32, 32, 32, 32) in (True, False)
    with pytest.raises(k.PdmkError):        # N not a multiple of 16
        k.gemm_geglu(rnd((64, 32), dev, dt), rnd((24, 32), dev, dt), gl, None, 64, 24, 32, 32, 32)


@pytest.mark.parametrize("da_dt,b_dt", [("bf16", "bf16"), ("bf16", "f32"), ("f32", "f32"), ("f32", "bf16")])
def test_mse_fused_forward_backward(dev, da_dt, b_dt):
    """pdmk_mse_fwd_bwd: loss value + gradient seed in one vectorised pass == the two scalar kernels == torch, with per-sample
    weights, strided rows, accumulate, forward-only and backward-only calls; odd shapes fall back to the scalar kernels."""
    from pdm import _pdmk as k
    torch.manual_seed(23)
    B, rows, cols, lda, ldb = 3, 50, 40, 48, 56
    a = rnd((B * rows, lda), dev, DT[da_dt])
    b = rnd((B * rows, ldb), dev, DT[b_dt])
    w = torch.rand(B, device=dev) + 0.5
    scale, gs = 1.0 / (B * rows * cols), 0.37
    d = a[:, :cols].float() - b[:, :cols].float()
    wr = w.repeat_interleave(rows)[:, None]
    ref_loss = float((d * d * wr).sum().double() * scale)
    out = torch.zeros(4, device=dev, dtype=torch.float64)
    da = torch.full((B * rows, cols), 2.0, device=dev, dtype=DT[da_dt])
    k.mse_fwd_bwd(a, b, w, out, 1, da, B, rows, cols, lda, ldb, cols, scale, gs, False)
    assert abs(float(out[1]) - ref_loss) <= 1e-5 * abs(ref_loss) and float(out[0]) == 0.0
    close(da, gs * wr * d, TOL[da_dt], "mse seed")
    k.mse_fwd_bwd(a, b, w, None, 0, da, B, rows, cols, lda, ldb, cols, scale, gs, True)          # backward only, accumulate
    close(da, 2 * gs * wr * d, TOL[da_dt] * 2, "mse seed accumulated")
    assert float(out[1]) == pytest.approx(ref_loss, rel=1e-5)
    k.mse_fwd_bwd(a, b, None, out, 2, None, B, rows, cols, lda, ldb, cols, scale, gs, False)      # forward only, no weights
    assert float(out[2]) == pytest.approx(float((d * d).sum().double() * scale), rel=1e-5)
    # a shape the vector kernel does not take (cols % 8 != 0) goes through the scalar pair with the same results
    out2 = torch.zeros(4, device=dev, dtype=torch.float64)
    da2 = torch.zeros(B * rows, 4, device=dev, dtype=DT[da_dt])
    k.mse_fwd_bwd(a, b, w, out2, 0, da2, B, rows, 4, lda, ldb, 4, scale, gs, False)
    d4 = a[:, :4].float() - b[:, :4].float()
    assert float(out2[0]) == pytest.approx(float((d4 * d4 * wr).sum().double() * scale), rel=1e-5)
    close(da2, gs * wr * d4, TOL[da_dt], "mse seed scalar fallback")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_groupnorm_bwd_extra_addend(dev, dn):
    """pdmk_groupnorm_bwd(add=...): dx (+)= GN-backward + add in one store (the residual branch's gradient of the same tensor),
    with and without accumulation into dx, strided operands."""
    from pdm import _pdmk as k
    torch.manual_seed(24)
    dt = DT[dn]
    B, HW, C, G = 2, 36, 64, 8
    gs = C // G
    x = rnd((B * HW, C), dev, dt)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y = torch.zeros_like(x)
    stats = torch.zeros(B, G, 2, device=dev)
    ws = k.groupnorm_ws(dev, B, G)
    k.groupnorm_fwd(x, y, gamma, beta, stats, ws, B, HW, C, C, C, G, gs, 1e-5, True)
    dy = rnd((B * HW, C), dev, dt)
    addbuf = rnd((B * HW, C + 16), dev, dt)
    add = addbuf[:, 8:8 + C]                                   # strided view, 16-byte aligned
    ref = torch.zeros_like(x)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    k.groupnorm_bwd(x, dy, ref, gamma, beta, stats, dg, db, ws, B, HW, C, C, C, C, G, gs, True, False)
    for acc in (False, True):
        dx = rnd((B * HW, C), dev, dt)
        base = dx.float().clone() if acc else 0.0
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        k.groupnorm_bwd(x, dy, dx, gamma, beta, stats, dg2, db2, ws, B, HW, C, C, C, C, G, gs, True, acc, add=add)
        close(dx, ref.float() + add.float() + base, TOL[dn] * 2, f"gn bwd + addend (acc={acc})")
        close(dg2, dg, 1e-5, "dgamma unchanged")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
@pytest.mark.parametrize("C,M", [(64, 37), (320, 300), (1280, 64)])
def test_layernorm_bwd_extra_addend(dev, dn, C, M):
    """pdmk_layernorm_bwd(add=...): dx (+)= LN-backward + add in one store - the residual stream's finished gradient
    (BasicTransformerBlock `attn_output + hidden_states`, blocks.py:705-867 backward) - with and without accumulation into dx,
    strided addend; the affine gradients do not change."""
    from pdm import _pdmk as k
    torch.manual_seed(25)
    dt = DT[dn]
    x = rnd((M, C), dev, dt) + 0.3
    gamma, beta = torch.randn(C, device=dev) * 0.3 + 1, torch.randn(C, device=dev) * 0.3
    y = torch.zeros_like(x)
    stats = torch.zeros(M, 2, device=dev)
    k.layernorm_fwd(x, y, gamma, beta, stats, M, C, C, C, 1e-5)
    dy = rnd((M, C), dev, dt)
    addbuf = rnd((M, C + 16), dev, dt)
    add = addbuf[:, 8:8 + C]
    ref = torch.zeros_like(x)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    k.layernorm_bwd(x, dy, ref, gamma, stats, dg, db, M, C, C, C, C, False)
    for acc in (False, True):
        dx = rnd((M, C), dev, dt)
        base = dx.float().clone() if acc else 0.0
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        k.layernorm_bwd(x, dy, dx, gamma, stats, dg2, db2, M, C, C, C, C, acc, add=add)
        close(dx, ref.float() + add.float() + base, TOL[dn] * 2, f"ln bwd + addend (acc={acc})")
        close(dg2, dg, 1e-5, "dgamma unchanged")
        close(db2, db, 1e-5, "dbeta unchanged")


def test_wgrad_group_of_a_transformer_block_matches_separate_launches(dev):
    """pdm._pdmk.wgrad_group: the eight Linear weight gradients of one transformer block (one reduction length, outputs from
    [224, 224] to [1792, 224], two of them with a bias gradient) as grouped launches with ONE split factor against the same
    gradients computed one by one (k.wgrad, each with its own plan) and against fp32 torch; a second group of another reduction
    length in the same call.  Same arithmetic up to the fp32 summation order."""
    from pdm import _pdmk as k
    torch.manual_seed(31)
    dt = torch.bfloat16
    P1, P2 = 4096, 1024
    shapes = [(672, 224, P1, True), (224, 224, P1, True), (224, 224, P1, False), (224, 224, P1, True), (1792, 224, P1, True),
              (224, 896, P1, True), (224, 224, P1, True), (224, 224, P1, False), (320, 320, P2, True), (320, 640, P2, False),
              (224, 96, P2, True)]      # (the engine pads pruned widths to multiples of 32: 224 = a 208-channel layer)
    items, refs, seps, q = [], [], [], k.SlabQueue()
    for No, Ki, P, has_bias in shapes:
        dy, x = rnd((P, No), dev, dt), rnd((P, Ki), dev, dt)
        dW = torch.randn(No, Ki, device=dev) * 0.1                 # gradients ACCUMULATE into the arena
        db = torch.randn(No, device=dev) * 0.1 if has_bias else None
        refs.append((dW + dy.float().t() @ x.float(), None if db is None else db + dy.float().sum(0)))
        dW2, db2 = dW.clone(), None if db is None else db.clone()
        k.wgrad(dy, x, dW2, No, Ki, P, No, Ki, colsum_out=db2)
        seps.append((dW2, db2))
        items.append((dy, x, dW, No, Ki, P, No, Ki, db, No * Ki * P))
    k.STATS.update(launches=0, grouped=0)
    k.wgrad_group(items, q)
    q.flush()
    torch.cuda.synchronize()
    assert k.STATS["launches"] >= 2                                 # one grouped call per reduction length
    for (dy, x, dW, No, Ki, P, *_rest), (rW, rb), (sW, sb), it in zip(items, refs, seps, items):
        close(dW, rW, 2e-3, f"grouped wgrad {No}x{Ki} P={P}")
        close(dW, sW, 1e-4, f"grouped vs separate wgrad {No}x{Ki} P={P}")
        if rb is not None:
            close(it[8], rb, 2e-3, "grouped bias gradient")
            close(it[8], sb, 1e-4, "grouped vs separate bias gradient")
    # the two 3x3 conv weight gradients of a ResBlock as one group (atomics for the splits), against the separate launches
    Bc, Hc = 2, 32
    Pc = Bc * Hc * Hc
    citems, cseps = [], []
    for Co, Ci in ((160, 64), (96, 160)):
        dy, x = rnd((Pc, Co), dev, dt), rnd((Pc, Ci), dev, dt)
        dW = torch.randn(Co, 9 * Ci, device=dev) * 0.1
        db = torch.randn(Co, device=dev) * 0.1
        conv = (Bc, Hc, Hc, Ci, Hc, Hc, 0, Ci)
        dW2, db2 = dW.clone(), db.clone()
        k.wgrad(dy, x, dW2, Co, 9 * Ci, Pc, Co, 0, b_mode=k.B_COLK_CONV, conv=conv, colsum_out=db2)
        cseps.append((dW2, db2))
        citems.append((dy, x, dW, Co, 9 * Ci, Pc, Co, 0, db, Co * 9 * Ci * Pc, k.B_COLK_CONV, conv))
    k.wgrad_group(citems, q)
    q.flush()
    torch.cuda.synchronize()
    for it, (sW, sb) in zip(citems, cseps):
        close(it[2], sW, 1e-4, "grouped vs separate conv weight gradient")
        close(it[8], sb, 1e-4, "grouped vs separate conv bias gradient")


@pytest.mark.parametrize("dn", ["f32", "bf16"])
def test_quantize_e4m3_matches_the_oracle_bit_for_bit(dev, dn):
    """pdmk_quantize_e4m3 (the "fp8_e4m3" attention precision of BASELINE.json configs[4]): in-place rounding to the e4m3fn value grid,
    bit-identical to the oracle's quant_e4m3 (itself equal to torch's float8_e4m3fn cast, tests/test_oracle_golden.py) on random
    values over five decades, ties, subnormals, the saturation edge and NaN; a length that leaves a scalar tail."""
    from pdm import _pdmk as k
    from pdm_ref import unet as ounet
    torch.manual_seed(8)
    dt = DT[dn]
    x = torch.cat([torch.randn(30001) * 3, torch.randn(3000) * 100, torch.randn(3000) * 0.01,
                   torch.tensor([0.0, -0.0, 448.0, 449.0, 1e6, -1e6, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 17.0, 19.0, -17.0,
                                 float("nan")])]).to(dt)
    want = ounet.quant_e4m3(x)
    got = x.clone().to(dev)
    k.quantize_e4m3_(got)
    torch.cuda.synchronize()
    got = got.cpu()
    nan = torch.isnan(want.float())
    assert torch.equal(torch.isnan(got.float()), nan)
    assert torch.equal(got.float()[~nan], want.float()[~nan])


def test_comm_handle_single_rank_allreduce(dev):
    """pdmk_comm_t (the C ABI's communicator handle): id -> create -> in-place fp32 all-reduce(sum) on a side stream ->
    destroy, on a world of one rank (all this one-GPU box can host: RCCL refuses two ranks on one device; the N-rank
    logic around it is covered by the gloo rehearsal tests)."""
    from pdm import _pdmk as k
    uid = k.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = k.Comm(uid, 0, 1)
    assert k._lib.pdmk_comm_world(comm._h) == 1
    x = torch.randn(1 << 20, device=dev)
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce_sum_(x)
        comm.all_reduce_sum_(x[1000:5000])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    comm.close()
    with pytest.raises(k.PdmkError):
        k.Comm(uid, 3, 2)                       # rank outside the world


def _unforce():
    for var in ("PDMK_RING_CFG", "PDMK_WGRAD_CFG"):
        os.environ.pop(var, None)


def _group_call(k, recs_fn, n_expected, cand, monkey_env):
    """Runs recs_fn() (which issues k.gemm calls) under a Recorder and sends the records through pdmk_gemm_group with the
    grouped candidate forced; returns how many problems went out in the multi-problem launch."""
    with k.Recorder() as r:
        recs_fn()
    assert len(r.recs) == n_expected and all(x.kind == "gemm" for x in r.recs)
    os.environ["PDMK_GROUP_CFG"] = str(cand)
    try:
        return k.gemm_group(r.recs)
    finally:
        del os.environ["PDMK_GROUP_CFG"]


@pytest.mark.parametrize("cand", [1, 3, 6, 7, 9, 12, 17, 19])
def test_gemm_group_linear_bit_equal_to_separate(dev, force_cfg, cand):
    """pdmk_gemm_group: 2, 3 and 4 independent Linear problems of DIFFERENT ragged shapes (bias / residual / accumulate /
    strided rows / fused GEGLU mixed) in one launch of ring candidate `cand` == the same problems launched one by one with
    that candidate, bit for bit; and vs fp32 math.  This is what runs teacher layer l and student layer l as one launch."""
    from pdm import _pdmk as k
    torch.manual_seed(cand)
    dt = torch.bfloat16
    probs = []
    for (M, N, K, bias, res, acc, geglu) in [(515, 320, 320, True, True, False, False), (200, 136, 96, False, False, True, False),
                                             (1000, 416, 64, True, False, False, True), (77, 64, 1024, True, True, False, False)]:
        A, W = rnd((M, K), dev, dt), rnd((N, K), dev, dt, 0.1)
        b = rnd((N,), dev, torch.float32) if bias else None
        R = rnd((M, N + 8), dev, dt) if res else None
        ld = (N // 2 if geglu else N) + 16
        C0 = rnd((M, ld), dev, dt)
        probs.append((M, N, K, A, W, b, R, acc, geglu, C0))

    def issue(outs, f_outs):
        for (M, N, K, A, W, b, R, acc, geglu, C0), C, F_ in zip(probs, outs, f_outs):
            if geglu:
                assert k.gemm_geglu(A, W, C[:, :N // 2], F_, M, N, K, K, K, bias=b)
            else:
                k.gemm(A, W, C[:, :N], M, N, K, K, K, C.stride(0), bias=b, R=R[:, :N] if R is not None else None,
                       ldr=R.stride(0) if R is not None else 0, accumulate=acc)
    for n in (2, 3, 4):
        force_cfg("PDMK_RING_CFG", cand)
        sep = [p[9].clone() for p in probs[:n]]
        sep_f = [torch.zeros(p[0], p[1], device=dev, dtype=dt) if p[8] else None for p in probs[:n]]
        pr = probs
        probs = pr[:n]
        issue(sep, sep_f)
        torch.cuda.synchronize()
        _unforce()
        grp = [p[9].clone() for p in probs]
        grp_f = [torch.zeros(p[0], p[1], device=dev, dtype=dt) if p[8] else None for p in probs]
        got = _group_call(k, lambda: issue(grp, grp_f), n, cand, None)
        torch.cuda.synchronize()
        probs = pr
        assert got == n, (cand, n, got)
        for a_, b_ in zip(sep, grp):
            assert torch.equal(a_, b_), (cand, n)
        for a_, b_ in zip(sep_f, grp_f):
            assert a_ is None or torch.equal(a_, b_)
    # and against fp32 math (first problem)
    M, N, K, A, W, b, R, acc, geglu, C0 = probs[0]
    ref = A.float() @ W.float().t() + b + R[:, :N].float()
    close(grp[0][:, :N].float(), ref, 2e-2, "group vs fp32")


@pytest.mark.parametrize("cand", [13, 14, 15, 16, 3, 12])
def test_gemm_group_conv_bit_equal_to_separate(dev, force_cfg, cand):
    """Grouped stride-1 3x3 convs (halo kernels 13-16; ring kernels 3 / 12 gather per tap): a dense and a pruned layer of one
    level, different channel counts, per-image time-embedding row / residual - one launch == separate launches."""
    from pdm import _pdmk as k
    torch.manual_seed(100 + cand)
    dt = torch.bfloat16
    B, H = 2, 16
    probs = []
    for (Ci, Co, rowvec, res) in [(64, 160, True, False), (96, 64, False, True), (32, 320, False, False)]:
        x, w = rnd((B * H * H, Ci), dev, dt), rnd((Co, 9 * Ci), dev, dt, 0.05)
        rv = rnd((B, Co), dev, torch.float32) if rowvec else None
        R = rnd((B * H * H, Co), dev, dt) if res else None
        probs.append((Ci, Co, x, w, rnd((Co,), dev, torch.float32), rv, R))

    def issue(outs):
        for (Ci, Co, x, w, b, rv, R), y in zip(probs, outs):
            k.gemm(x, w, y, B * H * H, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(B, H, H, Ci, H, H, 0, Ci), bias=b,
                   rowvec=rv, rows_per_b=H * H, ldrv=Co if rv is not None else 0, R=R, ldr=Co if R is not None else 0)
    force_cfg("PDMK_RING_CFG", cand)
    sep = [torch.zeros(B * H * H, p[1], device=dev, dtype=dt) for p in probs]
    issue(sep)
    torch.cuda.synchronize()
    _unforce()
    grp = [torch.zeros(B * H * H, p[1], device=dev, dtype=dt) for p in probs]
    got = _group_call(k, lambda: issue(grp), 3, cand, None)
    torch.cuda.synchronize()
    assert got == 3, (cand, got)
    for a_, b_ in zip(sep, grp):
        assert torch.equal(a_, b_), cand
    Ci, Co, x, w, b, rv, R = probs[0]
    ref = F.conv2d(x.float().view(B, H, H, Ci).permute(0, 3, 1, 2), w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2), b, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * H, Co) + rv.repeat_interleave(H * H, 0)
    close(grp[0].float(), ref, 2e-2, "group conv vs fp32")


@pytest.mark.parametrize("cand", [1, 2, 3, 4, 5, 6, 7])
def test_gemm_group_wgrad_matches_separate(dev, force_cfg, cand):
    """Grouped weight gradients (Linear: ring candidates 1-5; conv: 1-5 and the halo kernels 6 / 7), slab mode (plain stores:
    bit-equal to separate launches) with different split factors per problem, bias gradient fused."""
    from pdm import _pdmk as k
    torch.manual_seed(200 + cand)
    dt = torch.bfloat16
    conv = cand >= 6
    B, H = 2, 16
    P = B * H * H
    probs = []
    for (No, Ki, sk) in ([(64, 64, 2), (128, 96, 4), (160, 32, 1)] if conv else [(320, 320, 4), (160, 96, 2), (64, 1024, 3)]):
        dy, x = rnd((P, No), dev, dt), rnd((P, Ki), dev, dt)
        probs.append((No, Ki, sk, dy, x))

    def issue(outs, cs):
        for (No, Ki, sk, dy, x), ws, c in zip(probs, outs, cs):
            if conv:
                k.gemm(dy, x, ws, No, 9 * Ki, P, No, 0, 9 * Ki, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, out_f32=True, splitk=sk,
                       accumulate=2 if sk > 1 else 0, conv=(B, H, H, Ki, H, H, 0, Ki), dtype=k.BF16, colsum_out=c)
            else:
                k.gemm(dy, x, ws, No, Ki, P, No, Ki, Ki, a_mode=k.A_COLK, b_mode=k.B_COLK, out_f32=True, splitk=sk,
                       accumulate=2 if sk > 1 else 0, dtype=k.BF16, colsum_out=c)
    mk = lambda: ([torch.zeros(p[2] * p[0] * (9 if conv else 1) * p[1], device=dev) for p in probs],
                  [torch.zeros(p[0], device=dev) for p in probs])
    force_cfg("PDMK_WGRAD_CFG", cand)
    sep, sep_c = mk()
    issue(sep, sep_c)
    torch.cuda.synchronize()
    _unforce()
    grp, grp_c = mk()
    got = _group_call(k, lambda: issue(grp, grp_c), 3, cand, None)
    torch.cuda.synchronize()
    assert got == 3, (cand, got)
    for a_, b_ in zip(sep, grp):
        assert torch.equal(a_, b_), cand
    for a_, b_ in zip(sep_c, grp_c):
        assert torch.allclose(a_, b_, rtol=1e-5, atol=1e-5)          # bias gradient: fp32 atomics over the splits
    No, Ki, sk, dy, x = probs[0]
    got0 = grp[0].view(sk, No, -1).sum(0)
    if conv:
        xi = x.float().view(B, H, H, Ki).permute(0, 3, 1, 2)
        cols = torch.nn.functional.unfold(xi, 3, padding=1).view(B, Ki, 9, H * H).permute(0, 3, 2, 1).reshape(P, 9 * Ki)
        ref = dy.float().t() @ cols
    else:
        ref = dy.float().t() @ x.float()
    close(got0, ref, 2e-2, "group wgrad vs fp32")


@pytest.mark.parametrize("B,H,Ci,Co", [(2, 8, 64, 160), (2, 16, 96, 64), (1, 32, 32, 320), (3, 8, 160, 128)])
def test_upsample_conv_as_four_phase_convs(dev, B, H, Ci, Co):
    """Upsample2D = nearest x2 + 3x3 conv as four 2x2 phase convs on the low-resolution image
    (conv_mode 5..12, pdmk_up2_pack_weights / pdmk_up2_combine_wgrad): forward (one grouped launch of the four phases),
    input gradient and weight / bias gradient against F.interpolate + F.conv2d autograd in fp32 on the bf16-rounded
    operands.  Summing the 3x3 taps before the bf16 rounding is the only arithmetic difference: bf16 tolerance."""
    from pdm import _pdmk as k
    torch.manual_seed(B * 100 + H + Ci)
    dt = torch.bfloat16
    W = H
    assert k.conv_up2_supported(B, H, W, Ci, Co, dt)
    x = rnd((B * H * W, Ci), dev, dt)
    w3 = rnd((Co, 9 * Ci), dev, torch.float32, 0.05)            # packed [Co][9][Ci] fp32 master
    bias = rnd((Co,), dev, torch.float32)
    wp = torch.empty(4, Co, 4 * Ci, device=dev, dtype=dt)
    wpt = torch.empty(Ci, 16 * Co, device=dev, dtype=dt)          # [Ci][phase][tap][Co]
    k.up2_pack_weights(w3, wp, wpt, Co, Ci)
    geo = lambda m, ci, ld: (B, H, W, ci, H, W, m, ld)
    Ml, Mh = B * H * W, B * 4 * H * W
    y = torch.zeros(Mh, Co, device=dev, dtype=dt)
    with k.Recorder() as r:
        for p in range(4):
            k.gemm(x, wp[p], y, Ml, Co, 4 * Ci, 0, 4 * Ci, Co, a_mode=k.A_CONV, conv=geo(5 + p, Ci, Ci), bias=bias)
    k.gemm_group(r.recs)
    # reference: autograd on fp32 copies of the bf16 operands (weights: the fp32 master, as the 3x3 form would round them)
    xr = x.float().view(B, H, W, Ci).permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w3.view(Co, 3, 3, Ci).permute(0, 3, 1, 2).clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, br, padding=1)
    close(y.float(), yr.permute(0, 2, 3, 1).reshape(Mh, Co), 2e-2, "up2 forward")
    dy = rnd((Mh, Co), dev, dt)
    yr.backward(dy.float().view(B, 2 * H, 2 * W, Co).permute(0, 3, 1, 2))
    # input gradient: the four phases one after the other (modes 9..12, accumulating; a phase's weights are a column slice of
    # wpt), and all four as ONE problem (mode 13)
    dx = torch.zeros(Ml, Ci, device=dev, dtype=dt)
    for p in range(4):
        k.gemm(dy, wpt[:, p * 4 * Co:(p + 1) * 4 * Co], dx, Ml, Ci, 4 * Co, 0, 16 * Co, Ci, a_mode=k.A_CONV,
               conv=geo(9 + p, Co, Co), accumulate=p > 0)
    close(dx.float(), xr.grad.permute(0, 2, 3, 1).reshape(Ml, Ci), 3e-2, "up2 dgrad")
    dx1 = torch.zeros(Ml, Ci + 32, device=dev, dtype=dt)
    k.gemm(dy, wpt, dx1[:, :Ci], Ml, Ci, 16 * Co, 0, 16 * Co, Ci + 32, a_mode=k.A_CONV, conv=geo(13, Co, Co))
    close(dx1[:, :Ci].float(), xr.grad.permute(0, 2, 3, 1).reshape(Ml, Ci), 2e-2, "up2 dgrad (merged phases)")
    assert float(dx1[:, Ci:].abs().max()) == 0.0
    # weight / bias gradient
    dwp = torch.zeros(4, Co, 4 * Ci, device=dev)
    db = torch.zeros(Co, device=dev)
    sk = k.wgrad_plan(dy, x, Co, 4 * Ci, Ml, Co, 0, k.B_COLK_CONV, geo(5, Ci, Ci))
    with k.Recorder() as rw:
        for p in range(4):
            k.gemm(dy, x, dwp[p], Co, 4 * Ci, Ml, Co, 0, 4 * Ci, a_mode=k.A_COLK, b_mode=k.B_COLK_CONV, conv=geo(5 + p, Ci, Ci),
                   out_f32=True, splitk=sk, accumulate=(sk == 1), dtype=k.BF16, colsum_out=db)
    k.gemm_group(rw.recs)
    dw3 = torch.zeros(Co, 9 * Ci, device=dev)
    k.up2_combine_wgrad(dwp, dw3, Co, Ci)
    close(dw3, wr.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Ci), 2e-2, "up2 wgrad")
    close(db, br.grad, 2e-2, "up2 bias grad")
    # the packed phase weights are exact sums of the 3x3 taps (fp32 master -> one rounding)
    w9 = w3.view(Co, 3, 3, Ci)
    S = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
    for p in range(4):
        a, b = p >> 1, p & 1
        for dy_ in range(2):
            for dx_ in range(2):
                ref = sum(w9[:, ky, kx] for ky in S[(a, dy_)] for kx in S[(b, dx_)])
                got = wp[p].view(Co, 2, 2, Ci)[:, dy_, dx_]
                assert torch.equal(got, ref.to(dt)), (p, dy_, dx_)
                assert torch.equal(wpt.view(Ci, 4, 2, 2, Co)[:, p, 1 - dy_, 1 - dx_], got.t())


def test_tuner_scratch_has_a_canary_band(dev):
    """Regression guard of round 3's tuner overrun: the plan cache times candidates into a scratch output it sizes itself; a
    forward 2x2 phase conv (conv_mode 5..8) stores row m at a pixel of the 2H x 2W image, i.e. anywhere in 4 M rows, and the
    scratch was sized for M - a silent out-of-bounds write in ordinary runs, a memory fault under the profiler.  With
    PDMK_DEBUG_SCRATCH=1 every tuning pass gets a fresh scratch of exactly the computed size with a 1 MiB band of 0xA5
    behind it, checked after the timing launches (pdmk_debug_scratch_violations).  Tuning the four phases - one by one
    (tune_cfg) and as a group (tune_group) - must leave the band intact; PDMK_DEBUG_SCRATCH=2 sizes the scratch the OLD way
    for problems whose whole overrun fits the band, and the canary must fire (the check checks)."""
    import os
    from pdm import _pdmk as k
    B, H, Ci, Co = 1, 16, 32, 64
    W, dt = H, torch.bfloat16
    torch.manual_seed(5)
    x = rnd((B * H * W, Ci), dev, dt)
    w3 = rnd((Co, 9 * Ci), dev, torch.float32, 0.05)
    bias = rnd((Co,), dev, torch.float32)
    wp = torch.empty(4, Co, 4 * Ci, device=dev, dtype=dt)
    wpt = torch.empty(Ci, 16 * Co, device=dev, dtype=dt)
    k.up2_pack_weights(w3, wp, wpt, Co, Ci)
    geo = lambda m: (B, H, W, Ci, H, W, m, Ci)
    Ml, Mh = B * H * W, B * 4 * H * W
    yr = F.conv2d(F.interpolate(x.float().view(B, H, W, Ci).permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"),
                  w3.view(Co, 3, 3, Ci).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(Mh, Co)

    def run(grouped):
        y = torch.zeros(Mh, Co, device=dev, dtype=dt)
        if grouped:
            with k.Recorder() as r:
                for p in range(4):
                    k.gemm(x, wp[p], y, Ml, Co, 4 * Ci, 0, 4 * Ci, Co, a_mode=k.A_CONV, conv=geo(5 + p), bias=bias)
            k.gemm_group(r.recs)
        else:
            for p in range(4):
                k.gemm(x, wp[p], y, Ml, Co, 4 * Ci, 0, 4 * Ci, Co, a_mode=k.A_CONV, conv=geo(5 + p), bias=bias)
        torch.cuda.synchronize()
        close(y.float(), yr, 2e-2, "up2 forward under the scratch canary")
    saved = {v: os.environ.get(v) for v in ("PDMK_DEBUG_SCRATCH", "PDMK_RING_CFG", "PDMK_WGRAD_CFG")}
    for v in ("PDMK_RING_CFG", "PDMK_WGRAD_CFG"):
        os.environ.pop(v, None)
    try:
        os.environ["PDMK_DEBUG_SCRATCH"] = "1"
        k.plan_clear()
        v0 = k.debug_scratch_violations()
        run(False)                                     # tune_cfg of every phase
        assert k.plan_size() >= 4
        run(True)                                      # tune_group of the four
        assert k.debug_scratch_violations() == v0, "the tuner wrote behind its scratch"
        # positive control: size the scratch for M rows again - the canary must notice
        os.environ["PDMK_DEBUG_SCRATCH"] = "2"
        k.plan_clear()
        run(False)
        assert k.debug_scratch_violations() > v0, "the canary band did not catch a deliberate overrun"
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
        k.plan_clear()


@pytest.mark.parametrize("cand", [1, 4, 6, 8, 12, 17, 19, -1])
def test_gemm_fused_geglu_backward_epilogue(dev, force_cfg, cand):
    """PDMK_EPI_GEGLU_BWD: the input gradient of FeedForward's second Linear pushed through GEGLU's backward in the GEMM's
    epilogue equals - bit for bit - the plain input-gradient GEMM stored in bf16 followed by pdmk_geglu_bwd(layout 1), for ring
    tile shapes with ragged M / N / K; and fp32 autograd of hidden * gelu(gate) on the same operands."""
    from pdm import _pdmk as k
    if cand >= 0:
        force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(23)
    dt = torch.bfloat16
    for M, Fh, C in ((515, 352, 160), (200, 48, 96), (1000, 1296, 320)):
        dy, wt = rnd((M, C), dev, dt), rnd((Fh, C), dev, dt, C ** -0.5)        # wt rows = W^T: [F, C]
        pre = rnd((M, 2 * Fh), dev, dt)
        d = torch.zeros(M, Fh, device=dev, dtype=dt)
        k.gemm(dy, wt, d, M, Fh, C, C, C, Fh)
        ref = torch.zeros(M, 2 * Fh, device=dev, dtype=dt)
        k.geglu_bwd(pre, d, ref, M, Fh, 2 * Fh, Fh, 2 * Fh, layout=1)
        got = torch.full((M, 2 * Fh + 8), 3.0, device=dev, dtype=dt)[:, :2 * Fh]
        assert k.gemm_geglu_bwd(dy, wt, pre, got, M, Fh, C, C, C)
        assert torch.equal(got, ref), (M, Fh, C, (got.float() - ref.float()).abs().max().item())
        # fp32 autograd on the same (bf16-rounded) operands
        hg = pre.float().reshape(M, Fh // 8, 2, 8).clone().requires_grad_(True)
        (hg[:, :, 0] * F.gelu(hg[:, :, 1])).reshape(M, Fh).backward((dy.float() @ wt.float().t()).to(dt).float())
        close(got.float(), hg.grad.reshape(M, 2 * Fh), 2e-2, "fused geglu backward vs fp32")
    os.environ["PDMK_RING_CFG"] = "0"             # the K-step-32 kernels have no such epilogue: the library picks a ring shape itself
    assert k.gemm_geglu_bwd(rnd((64, 32), dev, dt), rnd((32, 32), dev, dt), rnd((64, 64), dev, dt),
                            torch.zeros(64, 64, device=dev, dtype=dt), 64, 32, 32, 32, 32) in (True, False)


@pytest.mark.parametrize("cand", [1, 2, 4, 6, 9, 10, 12, 13, 14, 15, 16, 17, 19, 21, -1])
def test_gemm_groupnorm_statistics_epilogue(dev, force_cfg, cand):
    """pdmk_gemm_args.colstat: the per-(image, column) sums and sums of squares of the STORED bf16 output come out of the GEMM's
    epilogue as 64-bit fixed-point integers (order-independent: bit-reproducible) (Linear with bias + residual, a conv with the time-embedding row vector, into an accumulator slice at a column
    offset, accumulating over two producers), for every ring tile shape, the halo-conv shapes and the tuned plan (21 = a
    row-block id: no such epilogue there, the library launches a ring shape itself); pdmk_groupnorm_apply_colstat on these sums
    = pdmk_groupnorm_fwd on the tensor (unet_2d_blocks.py ResnetBlock2D norm1 / norm2, blocks.py:322-380)."""
    from pdm import _pdmk as k
    if cand >= 0:
        force_cfg("PDMK_RING_CFG", cand)
    torch.manual_seed(41)
    dt = torch.bfloat16

    def sums(y, Bn, rows):
        yf = y.float().reshape(Bn, rows, -1)
        return yf.sum(1), (yf * yf).sum(1)

    def val(acc):                                   # fixed point, 30 fraction bits in two limbs (low, high) -> [B, 2, cols] values
        a = acc.double()
        return (a[:, 1::2] * 2.0 ** 32 + a[:, 0::2]) / 2.0 ** 30

    def check(acc, col0, y, Bn, rows, what):
        s1, s2 = sums(y, Bn, rows)
        n = y.shape[1]
        close(val(acc)[:, 0, col0:col0 + n], s1, 2e-4, what + " sum")
        close(val(acc)[:, 1, col0:col0 + n], s2, 2e-4, what + " sum of squares")

    # Linear (proj_out: bias + residual into a strided concat view), 3 images of 256 / 64 rows
    for Bn, rows, N, K in ((3, 256, 320, 160), (2, 64, 96, 320), (5, 320, 648, 96)):
        M = Bn * rows
        a, w, bias = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5), torch.randn(N, device=dev)
        res = rnd((M, N + 8), dev, dt)[:, 8:]
        ybuf = torch.zeros(M, N + 16, device=dev, dtype=dt)
        y = ybuf[:, 8:8 + N]
        acc = torch.zeros(Bn, 4, N + 24, device=dev, dtype=torch.int64)
        k.gemm(a, w, y, M, N, K, K, K, N + 16, bias=bias, R=res, ldr=N + 8, rows_per_b=rows, colstat=(acc, 16))
        ref = (a.float() @ w.float().t() + bias + res.float())
        close(y, ref, 2e-2, "linear with statistics")
        check(acc, 16, y, Bn, rows, f"linear B{Bn} rows{rows} N{N}")
        assert (acc[:, :, :16] == 0).all() and (acc[:, :, 16 + N:] == 0).all()
        # a second producer adds into the same accumulator columns (fp32 atomics): twice the sums
        k.gemm(a, w, y, M, N, K, K, K, N + 16, bias=bias, R=res, ldr=N + 8, rows_per_b=rows, colstat=(acc, 16))
        s1, _ = sums(y, Bn, rows)
        close(val(acc)[:, 0, 16:16 + N], 2 * s1, 2e-4, "two producers")
        # integer atomics: the same launch adds exactly the same numbers again, in whatever order the workgroups arrive
        acc2 = torch.zeros_like(acc)
        for _ in range(2):
            k.gemm(a, w, y, M, N, K, K, K, N + 16, bias=bias, R=res, ldr=N + 8, rows_per_b=rows, colstat=(acc2, 16))
        assert torch.equal(acc2, acc), "statistics must not depend on the arrival order of the workgroups"
    # 3x3 conv with the per-image row vector (ResnetBlock2D conv1 + time embedding), whole-row and 2D halo tiles
    for Bn, Hs, Ci, Co in ((2, 16, 64, 160), (1, 32, 32, 64), (3, 8, 96, 320), (2, 64, 32, 128)):
        x = rnd((Bn, Hs, Hs, Ci), dev, dt)
        w = rnd((Co, Ci, 3, 3), dev, dt, (9 * Ci) ** -0.5)
        bias, rv = torch.randn(Co, device=dev), torch.randn(Bn, Co, device=dev)
        M, rows = Bn * Hs * Hs, Hs * Hs
        y = torch.zeros(M, Co, device=dev, dtype=dt)
        acc = torch.zeros(Bn, 4, Co, device=dev, dtype=torch.int64)
        k.gemm(x, conv_w_pack(w), y, M, Co, 9 * Ci, 0, 9 * Ci, Co, a_mode=k.A_CONV, conv=(Bn, Hs, Hs, Ci, Hs, Hs, 0, Ci),
               bias=bias, rowvec=rv, rows_per_b=rows, colstat=(acc, 0))
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1) + rv[:, :, None, None]
        close(y, ref.permute(0, 2, 3, 1).reshape(M, Co), 2e-2, "conv with statistics")
        check(acc, 0, y, Bn, rows, f"conv B{Bn} {Hs}x{Hs} {Ci}->{Co}")
        # GroupNorm(+SiLU) from these sums == the two-pass GroupNorm of the same tensor
        G, gs = 32, Co // 32
        gamma, beta = torch.randn(Co, device=dev) * 0.3 + 1, torch.randn(Co, device=dev) * 0.3
        z0, z1 = torch.zeros_like(y), torch.zeros_like(y)
        st0, st1 = torch.zeros(Bn, G, 2, device=dev), torch.zeros(Bn, G, 2, device=dev)
        k.groupnorm_fwd(y, z0, gamma, beta, st0, k.groupnorm_ws(dev, Bn, G), Bn, rows, Co, Co, Co, G, gs, 1e-5, True)
        k.groupnorm_apply_colstat(y, z1, gamma, beta, st1, acc, 0, Bn, rows, Co, Co, Co, G, gs, 1e-5, True)
        close(st1, st0, 1e-4, "statistics from the epilogue vs the statistics pass")
        close(z1, z0, 1e-2, "groupnorm from epilogue statistics")
    # split-K producers: the statistics leave with the finish pass (pdmk_splitk_finish_colstat), slabs and ragged columns
    for Bn, rows, N, K, sk in ((2, 64, 1280, 640, 3), (4, 256, 328, 320, 2)):
        M = Bn * rows
        a, w, bias = rnd((M, K), dev, dt), rnd((N, K), dev, dt, K ** -0.5), torch.randn(N, device=dev)
        res, rv = rnd((M, N), dev, dt), torch.randn(Bn, N, device=dev)
        ws = torch.zeros(sk, M, N, device=dev)
        k.gemm(a, w, ws, M, N, K, K, K, N, out_f32=True, splitk=sk, accumulate=2)
        y0, y1 = torch.zeros(M, N, device=dev, dtype=dt), torch.zeros(M, N + 8, device=dev, dtype=dt)[:, :N]
        acc = torch.zeros(Bn, 4, N + 8, device=dev, dtype=torch.int64)
        k.splitk_finish(ws, y0, M, N, N, sk, bias=bias, rowvec=rv, R=res, ldr=N, rows_per_b=rows)
        k.splitk_finish(ws, y1, M, N, N + 8, sk, bias=bias, rowvec=rv, R=res, ldr=N, rows_per_b=rows, colstat=(acc, 8))
        assert torch.equal(y0, y1)
        check(acc, 8, y1, Bn, rows, f"split-K finish B{Bn} rows{rows} N{N}")
        assert (acc[:, :, :8] == 0).all()
    # activations far outside any sane range (a diverged run: 3e6, sums of squares of 1e15 per image) still give exact statistics
    Bn, rows, N, K = 2, 256, 64, 64
    a = (3.0e6 * torch.sign(torch.randn(Bn * rows, K, device=dev))).to(dt)
    w = torch.eye(N, K, device=dev).to(dt)
    y = torch.zeros(Bn * rows, N, device=dev, dtype=dt)
    acc = torch.zeros(Bn, 4, N, device=dev, dtype=torch.int64)
    k.gemm(a, w, y, Bn * rows, N, K, K, K, N, rows_per_b=rows, colstat=(acc, 0))
    check(acc, 0, y, Bn, rows, "huge activations")
    # a NaN / inf in the tensor must give non-finite statistics for ITS (image, group) - as the statistics pass does - and leave every
    # other group exact: the non-finite partial sets a flag bit in the column's sum-of-squares limb (common.h cs_add), whatever else
    # is added to that column before or after
    Bn, rows, N, K = 2, 128, 64, 64
    for poison in (float("nan"), float("inf")):
        a = rnd((Bn * rows, K), dev, dt)
        clean = torch.zeros(Bn * rows, N, device=dev, dtype=dt)
        res = clean.clone()
        res[rows + 5, 9] = poison                                     # ONE element: image 1, column 9 -> group 9 // 2 = 4
        y, y2 = torch.zeros(Bn * rows, N, device=dev, dtype=dt), torch.zeros(Bn * rows, N, device=dev, dtype=dt)
        G, gs = 32, 2
        gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        z = torch.zeros_like(y)
        st = torch.zeros(Bn, G, 2, device=dev)
        acc1 = torch.zeros(Bn, 4, N, device=dev, dtype=torch.int64)
        # finite additions to the same columns before and after the poisoned launch: the flag must survive both
        for R_ in (clean, res, clean):
            k.gemm(a, w, y if R_ is res else y2, Bn * rows, N, K, K, K, N, R=R_, ldr=N, rows_per_b=rows, colstat=(acc1, 0))
        assert not torch.isfinite(y[rows + 5, 9]) and torch.isfinite(y2).all()
        k.groupnorm_apply_colstat(y, z, gamma, beta, st, acc1, 0, Bn, rows, N, N, N, G, gs, 1e-5, False)
        bad = ~torch.isfinite(st).all(dim=2)
        want = torch.zeros(Bn, G, dtype=torch.bool, device=dev)
        want[1, 4] = True
        assert torch.equal(bad, want), (poison, bad.nonzero().tolist())
        st0 = torch.zeros_like(st)
        k.groupnorm_fwd(y2, torch.zeros_like(y), gamma, beta, st0, k.groupnorm_ws(dev, Bn, G), Bn, rows, N, N, N, G, gs, 1e-5, False)
        good = ~want
        # (three launches were added with n counted once: the clean groups' "mean" is 3 x the mean of one tensor)
        close(st[good][:, 0], 3 * st0[good][:, 0], 1e-4, "sums of the clean groups beside a non-finite one")
    # shapes the epilogue does not take are refused (-1), never silently skipped
    y = torch.zeros(96, 64, device=dev, dtype=dt)
    with pytest.raises(k.PdmkError):
        k.gemm(rnd((96, 32), dev, dt), rnd((64, 32), dev, dt), y, 96, 64, 32, 32, 32, 64, rows_per_b=48,
               colstat=(torch.zeros(2, 4, 64, device=dev, dtype=torch.int64), 0))


@pytest.mark.parametrize("dn", ["bf16", "f32"])
def test_transpose_tiles_vector_and_scalar_paths(dev, dn):
    """pdmk_transpose_tiles (the dgrad weight copies, params.py refresh_wt): 64x64-tile table records over strided sources and
    destinations - the 16-byte path (offsets, row strides and extents in whole chunks; ragged last tiles in both directions), the
    element-wise path (odd strides / offsets) and a conv-style job (src [co][9][ci] tap slice -> dst [ci][9][co] flipped tap)."""
    import numpy as np
    from pdm import _pdmk as k
    torch.manual_seed(3)
    dt = DT[dn]

    def table(jobs):
        recs = []
        for so, do, rows, cols, sld, dld in jobs:
            rr, cc = np.meshgrid(np.arange(0, rows, 64), np.arange(0, cols, 64), indexing="ij")
            t = np.zeros((rr.size, 12), dtype=np.int64)
            t[:, 0], t[:, 2] = so, do
            t[:, 4], t[:, 5], t[:, 6], t[:, 7] = rows, cols, sld, dld
            t[:, 8], t[:, 9] = rr.ravel(), cc.ravel()
            recs.append(t)
        tab = np.concatenate(recs, 0).astype(np.int32)
        return torch.from_numpy(tab).to(dev), tab.shape[0]

    for rows, cols, sld, dld, so, do in ((200, 136, 136, 200, 0, 0), (64, 64, 64, 64, 0, 0), (328, 72, 80, 336, 16, 64),
                                         (50, 37, 37, 50, 0, 0), (72, 40, 41, 72, 3, 0), (130, 320, 320, 136, 8, 24)):
        src = rnd((so + rows * sld + 8,), dev, dt)
        dst = torch.full((do + cols * dld + 8,), 7.0, device=dev, dtype=dt)
        tab, n = table([(so, do, rows, cols, sld, dld)])
        k.transpose_tiles(src, dst, tab, n)
        s2 = src[so:so + rows * sld].view(rows, sld)[:, :cols]
        d2 = dst[do:do + cols * dld].view(cols, dld)
        assert torch.equal(d2[:, :rows], s2.t()), (rows, cols, sld, dld, so, do)
        assert (d2[:, rows:] == 7.0).all() and (dst[:do] == 7.0).all()
    # conv weight: src [co][9][ci] -> dst [ci][9 flipped][co], nine jobs in one launch
    co, ci = 96, 40
    w = rnd((co, 9, ci), dev, dt)
    wt = torch.zeros(ci, 9, co, device=dev, dtype=dt)
    tab, n = table([(t_ * ci, (8 - t_) * co, co, ci, 9 * ci, 9 * co) for t_ in range(9)])
    k.transpose_tiles(w, wt, tab, n)
    assert torch.equal(wt, w.flip(1).permute(2, 1, 0).contiguous())
