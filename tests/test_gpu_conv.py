"""Kernel-level parity of the lip-CNN building blocks (csrc/conv.hip, conv_direct.hip) through the C ABI against a plain
PyTorch fp32/fp64 reference on the CPU (torch.nn.functional.conv2d + autograd), with TensorFlow's SAME geometry.
Tolerance: 2e-5 of the tensor's magnitude (fp32 accumulation order differs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2, total - total // 2


def _ref_conv(x, w, b, s):
    """x [N,H,W,C] float64, w [k,k,Ci,Co]: TF conv2d(padding='SAME')."""
    k = w.shape[0]
    Ho, pt, pb = _same(x.shape[1], k, s)
    Wo, pl, pr = _same(x.shape[2], k, s)
    xt = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = torch.nn.functional.conv2d(xt, w.permute(3, 2, 0, 1), b, stride=s)
    return y.permute(0, 2, 3, 1)


def _close(a, b, tol=2e-5):
    b = np.asarray(b)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("N,H,Ci,Co,s", [(3, 12, 3, 8, 1), (2, 12, 8, 8, 1), (2, 12, 8, 16, 2), (2, 9, 16, 16, 1), (5, 9, 16, 16, 2), (2, 7, 4, 4, 2),
                                        (3, 18, 16, 32, 2), (19, 9, 32, 32, 1)])
def test_direct_conv3x3_forward_and_gradients(N, H, Ci, Co, s):
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(H * 100 + Ci * 10 + Co + s)
    W = H
    x = torch.tensor(rng.standard_normal((N, H, W, Ci)), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(rng.standard_normal((3, 3, Ci, Co)) * 0.3, dtype=torch.float64, requires_grad=True)
    b = torch.tensor(rng.standard_normal(Co), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(x, w, b, s)
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
    (y * dy).sum().backward()
    Ho, pt, _ = _same(H, 3, s)
    Wo, pl, _ = _same(W, 3, s)
    assert ops.conv3x3_supported(Ci, Co, H, W)
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    yd = torch.zeros(N, Ho, Wo, Co, device="cuda")
    ops.conv3x3(xd, wd, bd, yd, N, H, W, Ci, Co, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    assert _close(yd.cpu().numpy(), y.detach().numpy())
    # weight gradient accumulates into dw (beta = 1): start from a known offset
    dw = torch.full((3, 3, Ci, Co), 0.5, device="cuda")
    scratch = torch.empty(1 << 20, device="cuda")
    ops.conv3x3_bwd_weight(xd, dyd, dw, N, H, W, Ci, Co, s, pt, pl, Ho, Wo, scratch)
    torch.cuda.synchronize()
    assert _close(dw.cpu().numpy() - 0.5, w.grad.numpy(), 5e-5)
    if Ci % 4 == 0:
        dx = torch.full((N, H, W, Ci), 0.25, device="cuda")
        if s == 1:
            ops.conv3x3(dyd, wd, None, dx, N, Ho, Wo, Co, Ci, 1, 1, 1, H, W, flip=1, beta=1.0)
        else:
            ops.conv3x3_bwd_data_s2(dyd, wd, dx, N, H, W, Ci, Co, pt, pl, Ho, Wo, beta=1.0)
        torch.cuda.synchronize()
        assert _close(dx.cpu().numpy() - 0.25, x.grad.numpy(), 5e-5)


@pytest.mark.parametrize("N,H,Ci,Co,k,s", [(2, 9, 32, 64, 3, 2), (3, 10, 16, 32, 1, 2), (2, 5, 64, 64, 3, 1)])
def test_im2col_gemm_conv_and_col2im(N, H, Ci, Co, k, s):
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(7 + H + Ci)
    W = H
    x = torch.tensor(rng.standard_normal((N, H, W, Ci)), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(rng.standard_normal((k, k, Ci, Co)) * 0.2, dtype=torch.float64)
    y = _ref_conv(x, w, None, s)
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
    (y * dy).sum().backward()
    Ho, pt, _ = _same(H, k, s)
    Wo, pl, _ = _same(W, k, s)
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    xd, wd, dyd = dev(x), dev(w), dev(dy)
    rows, K = N * Ho * Wo, k * k * Ci
    col = torch.zeros(rows, K, device="cuda")
    ops.im2col(xd, col, N, H, W, Ci, k, k, s, pt, pl, Ho, Wo)
    yd = torch.zeros(rows, Co, device="cuda")
    ops.gemm(ops.mat(col, K), ops.mat(wd, Co), ops.mat(yd, Co), rows, Co, K, splitk=1)
    torch.cuda.synchronize()
    assert _close(yd.cpu().numpy().reshape(y.shape), y.detach().numpy())
    dcol = torch.zeros(rows, K, device="cuda")
    ops.gemm(ops.mat(dyd, Co), ops.mat(wd, Co), ops.mat(dcol, K), rows, K, Co, trans_b=1, splitk=1)
    dx = torch.zeros(N, H, W, Ci, device="cuda")
    ops.col2im(dcol, dx, N, H, W, Ci, k, k, s, pt, pl, Ho, Wo)
    torch.cuda.synchronize()
    assert _close(dx.cpu().numpy(), x.grad.numpy(), 5e-5)


@pytest.mark.parametrize("rows,F,relu", [(700, 8, 1), (5000, 16, 1), (333, 64, 0), (130000, 8, 1), (6000, 64, 0), (4100, 12, 1), (40000, 32, 1)])
def test_batchnorm_forward_backward(rows, F, relu):
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(rows + F)
    x = torch.tensor(rng.standard_normal((rows, F)) * 2 + 0.5, dtype=torch.float64, requires_grad=True)
    g = torch.tensor(rng.uniform(0.5, 1.5, F), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(rng.standard_normal(F) * 0.3, dtype=torch.float64, requires_grad=True)
    eps, mom = 1e-5, 0.98
    mean = x.mean(0)
    var = ((x - mean) ** 2).mean(0)
    y = (x - mean) * torch.rsqrt(var + eps) * g + b
    if relu:
        y = torch.relu(y)
    dy = torch.tensor(rng.standard_normal((rows, F)), dtype=torch.float64)
    (y * dy).sum().backward()
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    xd, gd, bd, dyd = dev(x), dev(g), dev(b), dev(dy)
    yd = torch.zeros(rows, F, device="cuda")
    mm, mv = torch.zeros(F, device="cuda"), torch.ones(F, device="cuda")
    sm, si = torch.zeros(F, device="cuda"), torch.zeros(F, device="cuda")
    scratch = torch.empty(1 << 20, device="cuda")
    ops.batchnorm_fwd_ex(xd, yd, rows, F, gd, bd, mm, mv, sm, si, True, eps, mom, relu, scratch)
    torch.cuda.synchronize()
    assert _close(yd.cpu().numpy(), y.detach().numpy(), 5e-5)
    assert _close(mm.cpu().numpy(), (1 - mom) * mean.detach().numpy(), 5e-5)
    assert _close(mv.cpu().numpy(), mom + (1 - mom) * var.detach().numpy() * rows / (rows - 1), 5e-5)
    dx, dg, db = torch.zeros(rows, F, device="cuda"), torch.zeros(F, device="cuda"), torch.zeros(F, device="cuda")
    ops.batchnorm_bwd(xd, dyd, gd, bd, sm, si, dx, dg, db, rows, F, relu, scratch)
    torch.cuda.synchronize()
    assert _close(dx.cpu().numpy(), x.grad.numpy(), 2e-4)
    assert _close(dg.cpu().numpy(), g.grad.numpy(), 2e-4) and _close(db.cpu().numpy(), b.grad.numpy(), 2e-4)


# ------------------------------------------------------------------------------------------------
# descriptor API of csrc/conv_mfma.hip: k = 1 / 3, stride 1 / 2, up to 64 channels (tap-group launches), BN-ReLU applied by the loader,
# residual + BN statistics in the epilogue, bias gradient from the weight-gradient pass.
@pytest.mark.parametrize("N,H,Ci,Co,k,s,bn,res", [
    (3, 12, 3, 8, 3, 1, False, False), (5, 12, 8, 8, 3, 1, True, True), (4, 12, 8, 16, 3, 2, True, False), (4, 12, 8, 16, 1, 2, False, False),
    (7, 9, 16, 32, 1, 2, False, False), (6, 9, 32, 64, 3, 2, True, False), (6, 9, 32, 64, 1, 2, False, False), (9, 5, 64, 64, 3, 1, True, True),
    (2, 36, 8, 8, 3, 1, True, True), (3, 18, 16, 16, 3, 1, True, False), (21, 5, 64, 64, 3, 1, False, False),
    # pixel-pair rows (8 destination channels, even width) next to the odd width that cannot pair; stride-2 data gradients with all four
    # parity classes in one launch on even and odd maps (partial 2x2 cells), 8 and 16 channels
    (3, 7, 8, 8, 3, 1, False, False), (3, 10, 8, 8, 3, 1, False, True), (5, 9, 16, 32, 3, 2, False, False), (4, 11, 8, 16, 3, 2, False, False),
    (3, 18, 16, 32, 3, 2, False, False), (70, 6, 8, 16, 3, 2, False, False),
    # projection shortcuts on odd maps, many frames (several pixel tiles per workgroup), 4 -> 4 and 64 -> 64 channels
    (5, 11, 8, 16, 1, 2, False, False), (300, 18, 16, 32, 1, 2, False, False), (33, 5, 64, 64, 1, 2, False, False), (9, 7, 4, 4, 1, 2, False, False)])
def test_conv_desc_forward_and_gradients(N, H, Ci, Co, k, s, bn, res):
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(H * 100 + Ci * 10 + Co + s + k)
    W = H
    t64 = lambda a, g=True: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    x = t64(rng.standard_normal((N, H, W, Ci)))
    w = t64(rng.standard_normal((k, k, Ci, Co)) * 0.3)
    b = t64(rng.standard_normal(Co))
    sc, sh = t64(rng.uniform(0.5, 1.5, Ci), False), t64(rng.standard_normal(Ci) * 0.3, False)
    xin = torch.relu(x * sc + sh) if bn else x
    y = _ref_conv(xin, w, b, s)
    r = t64(rng.standard_normal(tuple(y.shape)), False)
    rsc, rsh = t64(rng.uniform(0.5, 1.5, Co), False), t64(rng.standard_normal(Co) * 0.3, False)
    if res:
        y = y + torch.relu(r * rsc + rsh)
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
    (y * dy).sum().backward(inputs=[x, w, b] if not bn else [w, b])
    Ho, pt, _ = _same(H, k, s)
    Wo, pl, _ = _same(W, k, s)
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    bnv = (dev(sc), dev(sh)) if bn else None
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, pt if k == 3 else 0, pl if k == 3 else 0, Ho, Wo, bn=bnv)
    assert ops.conv_supported(d)
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    yd = torch.full((N, Ho, Wo, Co), 7.0, device="cuda")
    stats = torch.zeros(512 * 2 * Co, device="cuda")
    n = ops.conv_fwd(d, xd, wd, bd, yd, dev(r) if res else None, (dev(rsc), dev(rsh)) if res else None, stats)
    torch.cuda.synchronize()
    yr = y.detach().numpy()
    assert _close(yd.cpu().numpy(), yr)
    part = stats[:n * 2 * Co].view(n, 2, Co).double().sum(0).cpu().numpy()
    assert np.abs(part[0] - yr.sum((0, 1, 2))).max() <= 1e-4 * max(1.0, np.abs(yr).sum((0, 1, 2)).max())
    assert np.abs(part[1] - (yr ** 2).sum((0, 1, 2))).max() <= 1e-4 * (yr ** 2).sum((0, 1, 2)).max()
    if k == 1 and s == 2 and not res:
        # the projection shortcuts of the network run without statistics and, in one layout, without a bias gradient
        y2 = torch.full((N, Ho, Wo, Co), -3.0, device="cuda")
        ops.conv_fwd(d, xd, wd, bd, y2, None, None, None)
        torch.cuda.synchronize()
        assert _close(y2.cpu().numpy(), yr)
        dw2 = torch.full((k, k, Ci, Co), 0.5, device="cuda")
        ops.conv_bwd_weight(d, xd, dyd, dw2, None, torch.empty(1 << 22, device="cuda"))     # (without the bias gradient)
        torch.cuda.synchronize()
        assert _close(dw2.cpu().numpy() - 0.5, w.grad.numpy(), 5e-5)
    dw = torch.full((k, k, Ci, Co), 0.5, device="cuda")
    db = torch.full((Co,), -0.25, device="cuda")
    scratch = torch.empty(1 << 22, device="cuda")
    ops.conv_bwd_weight(d, xd, dyd, dw, db, scratch)
    torch.cuda.synchronize()
    assert _close(dw.cpu().numpy() - 0.5, w.grad.numpy(), 5e-5)
    assert _close(db.cpu().numpy() + 0.25, b.grad.numpy(), 5e-5)
    if Ci % 4 == 0 and not bn:
        dx = torch.full((N, H, W, Ci), 0.25, device="cuda")
        ops.conv_bwd_data(d, dyd, wd, dx, beta=1.0)
        torch.cuda.synchronize()
        assert _close(dx.cpu().numpy() - 0.25, x.grad.numpy(), 5e-5)
    elif Ci % 4 == 0:
        # the data gradient is with respect to the NORMALISED map (the batch-norm backward takes it from there)
        xn = xin.detach().clone().requires_grad_(True)
        (_ref_conv(xn, w.detach(), b.detach(), s) * dy).sum().backward()
        dx = torch.zeros((N, H, W, Ci), device="cuda")
        if k == 1 and s == 2:
            ops.conv_bwd_data(d, dyd, wd, dx, beta=1.0)
        else:
            dx.fill_(3.0)
            ops.conv_bwd_data(d, dyd, wd, dx, beta=0.0)
        torch.cuda.synchronize()
        assert _close(dx.cpu().numpy(), xn.grad.numpy(), 5e-5)


@pytest.mark.parametrize("N,H,Ci,Co,s,acc", [(5, 12, 8, 8, 1, True), (3, 36, 8, 8, 1, False), (4, 12, 8, 16, 2, False), (5, 9, 16, 32, 2, False),
                                             (6, 9, 16, 16, 1, True), (9, 5, 64, 64, 1, False), (4, 11, 8, 16, 2, False), (7, 9, 32, 32, 1, False),
                                             (300, 6, 8, 8, 1, True)])
def test_data_gradient_with_fused_batchnorm_backward(N, H, Ci, Co, s, acc):
    """avsr_conv_bwd_data_bn + avsr_bn_bwd_finalize + avsr_bn_bwd_apply against torch autograd (fp64) through
    x -> batch_norm(training statistics) -> relu -> conv (avsr/video.py:4-14, :57-88), with an optional second consumer of the
    normalised map (a residual connection) whose gradient arrives as `acc` and is read in place."""
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(N * 1000 + H * 10 + Ci + Co + s)
    W, k, eps = H, 3, 1e-5
    t64 = lambda a, g=True: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    x = t64(rng.standard_normal((N, H, W, Ci)) * 1.5 + 0.2)
    gamma, beta = t64(rng.uniform(0.5, 1.5, Ci)), t64(rng.standard_normal(Ci) * 0.3)
    w = t64(rng.standard_normal((k, k, Ci, Co)) * 0.3, False)
    mean = x.mean((0, 1, 2))
    var = ((x - mean) ** 2).mean((0, 1, 2))
    invstd = torch.rsqrt(var + eps)
    y = torch.relu((x - mean) * invstd * gamma + beta)
    out = _ref_conv(y, w, None, s)
    dy = torch.tensor(rng.standard_normal(tuple(out.shape)), dtype=torch.float64)
    racc = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float64)
    loss = (out * dy).sum() + ((y * racc).sum() if acc else 0.0)
    loss.backward()
    Ho, pt, _ = _same(H, k, s)
    Wo, pl, _ = _same(W, k, s)
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    scale = (gamma * invstd).detach()
    shift = (beta - mean * gamma * invstd).detach()
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, pt, pl, Ho, Wo)
    assert ops.conv_supported(d) and ops.conv_bwd_data_bn_supported(d)
    xd, wd, dyd = dev(x), dev(w), dev(dy)
    dz = torch.full((N, H, W, Ci), 3.0, device="cuda")
    stats = torch.zeros(512 * 2 * Ci, device="cuda")
    n = ops.conv_bwd_data_bn(d, dyd, wd, dz, beta=1.0 if acc else 0.0, acc=dev(racc) if acc else None, bn_x=xd, bn=(dev(scale), dev(shift)),
                             stats=stats)
    assert n > 0
    k3 = torch.zeros(3 * Ci, device="cuda")
    dg, db = torch.full((Ci,), 9.0, device="cuda"), torch.full((Ci,), 9.0, device="cuda")
    ops.bn_bwd_finalize(stats, n, Ci, N * H * W, dev(mean), dev(invstd), dev(gamma), dg, db, k3, grad_beta=0.0)
    dx = torch.full((N, H, W, Ci), 0.5, device="cuda")
    ops.bn_bwd_apply(dz, xd, k3, dx, N * H * W, Ci, beta=1.0)
    torch.cuda.synchronize()
    assert _close(dg.cpu().numpy(), gamma.grad.numpy(), 2e-4) and _close(db.cpu().numpy(), beta.grad.numpy(), 2e-4)
    assert _close(dx.cpu().numpy() - 0.5, x.grad.numpy(), 2e-4)
    # the masked gradient itself: zero exactly where the normalised map is not positive
    yn = y.detach().numpy()
    assert not np.any(dz.cpu().numpy()[yn <= 0])


@pytest.mark.parametrize("rows,Cn,lazy,alias", [(81 * 7, 32, True, True), (1, 4, True, False), (4099, 8, False, True), (300 * 25, 64, True, False),
                                                (513 * 9, 12, False, False), (37, 1024, True, True)])
def test_batchnorm_backward_stage_one_as_its_own_pass(rows, Cn, lazy, alias):
    """avsr_bn_bwd_stage1 (the ReLU mask + the partial sums a single-launch data gradient emits from its epilogue, as a pass of its own) +
    avsr_bn_bwd_finalize + avsr_bn_bwd_apply against torch autograd (fp64) through x -> batch_norm -> relu (avsr/video.py:4-14)."""
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(rows + Cn)
    eps = 1e-5
    t64 = lambda a, g=True: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    x = t64(rng.standard_normal((rows, Cn)) * 1.5 + 0.2)
    gamma, beta = t64(rng.uniform(0.5, 1.5, Cn)), t64(rng.standard_normal(Cn) * 0.3)
    mean = x.mean(0)
    var = ((x - mean) ** 2).mean(0)
    invstd = torch.rsqrt(var + eps)
    dev = lambda t: t.detach().to(torch.float32).cuda().contiguous()
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    sc32, sh32, y32 = f32(gamma * invstd), f32(beta - mean * gamma * invstd), f32(torch.relu((x - mean) * invstd * gamma + beta))
    # the mask the kernel takes, exactly: the sign of fma(x, scale, shift) in fp32 is the sign of the exact value, which fp64 holds
    mask = (f32(x).astype(np.float64) * sc32 + sh32 > 0) if lazy else (y32 > 0)
    y = ((x - mean) * invstd * gamma + beta) * torch.tensor(mask.astype(np.float64))
    dy = torch.tensor(rng.standard_normal((rows, Cn)), dtype=torch.float64)
    (y * dy).sum().backward()
    xd, dyd = dev(x), dev(dy)
    expect_dz = np.where(mask, f32(dy), np.float32(0.0))
    dz = dyd if alias else torch.full((rows, Cn), 3.0, device="cuda")
    part = torch.full((512 * 2 * Cn,), 7.0, device="cuda")
    if lazy:
        n = ops.bn_bwd_stage1(dyd, xd, dz, rows, Cn, part, scale=torch.tensor(sc32).cuda(), shift=torch.tensor(sh32).cuda())
    else:
        n = ops.bn_bwd_stage1(dyd, xd, dz, rows, Cn, part, y=torch.tensor(y32).cuda())
    assert 0 < n <= 512
    k3 = torch.zeros(3 * Cn, device="cuda")
    dg, db = torch.full((Cn,), 9.0, device="cuda"), torch.full((Cn,), 9.0, device="cuda")
    ops.bn_bwd_finalize(part, n, Cn, rows, dev(mean), dev(invstd), dev(gamma), dg, db, k3, grad_beta=0.0)
    dx = torch.zeros((rows, Cn), device="cuda")
    ops.bn_bwd_apply(dz, xd, k3, dx, rows, Cn)
    torch.cuda.synchronize()
    assert np.array_equal(dz.cpu().numpy(), expect_dz)
    assert _close(dg.cpu().numpy(), gamma.grad.numpy(), 2e-4) and _close(db.cpu().numpy(), beta.grad.numpy(), 2e-4)
    assert _close(dx.cpu().numpy(), x.grad.numpy(), 2e-4)


@pytest.mark.parametrize("N,H,Ci,Co,st", [(5, 12, 3, 8, 1), (3, 36, 3, 8, 1), (70, 10, 3, 8, 1), (3, 36, 8, 8, 1), (7, 10, 8, 8, 1), (4, 36, 8, 16, 2),
                                           (5, 18, 16, 32, 2), (6, 9, 32, 64, 2), (9, 11, 8, 16, 2)])
def test_weight_gradient_with_the_batchnorm_backward_in_its_operand_fetch(N, H, Ci, Co, st):
    """avsr_conv_bwd_weight_bn (round 5): the gradient of the convolution's output, dx = k1*dz + k2*y + k3 (avsr_bn_bwd_finalize's
    vectors), evaluated while the weight-gradient kernel fetches its operand -- and, for layers with a data gradient, stored by the lane
    that fetched it -- against avsr_bn_bwd_apply + avsr_conv_bwd_weight on the materialised dx: weight AND bias gradient AND the stored
    map, including maps whose last 16-position chunk of a frame is partial, every kernel form the lip CNN's layers take (pixel-pair
    8-channel, one / two / four column tiles, the row-split form), and the BN-ReLU loader on the input side."""
    from avsr_tf1_amd import ops
    rng = np.random.default_rng(N + H + Ci)
    dev = lambda a: torch.tensor(a, dtype=torch.float32).cuda().contiguous()
    Ho, pt, _ = _same(H, 3, st)
    x = dev(rng.standard_normal((N, H, H, Ci)))
    y = dev(rng.standard_normal((N, Ho, Ho, Co)))
    dz = dev(rng.standard_normal((N, Ho, Ho, Co)))
    k = dev(np.concatenate([rng.uniform(0.5, 1.5, Co), rng.standard_normal(Co) * 0.3, rng.standard_normal(Co) * 0.2]))
    bnv = (dev(rng.uniform(0.5, 1.5, Ci)), dev(rng.standard_normal(Ci) * 0.3)) if Ci % 4 == 0 else None
    d = ops.conv_desc(N, H, H, Ci, Co, 3, st, pt, pt, Ho, Ho, bn=bnv)
    assert ops.conv_bwd_weight_bn_supported(d)
    dx = torch.zeros_like(y)
    ops.bn_bwd_apply(dz, y, k, dx, N * Ho * Ho, Co)
    scratch = torch.empty(1 << 23, device="cuda")
    dw0, db0 = torch.full((3, 3, Ci, Co), 0.5, device="cuda"), torch.full((Co,), -0.25, device="cuda")
    ops.conv_bwd_weight(d, x, dx, dw0, db0, scratch)
    for store in ((False, True) if Ci % 4 == 0 else (False,)):
        dw1, db1 = torch.full((3, 3, Ci, Co), 0.5, device="cuda"), torch.full((Co,), -0.25, device="cuda")
        out = torch.full_like(y, 9.0) if store else None
        ops.conv_bwd_weight_bn(d, x, dz, y, k, dw1, db1, torch.empty(1 << 23, device="cuda"), dx_out=out)
        torch.cuda.synchronize()
        assert _close(dw1.cpu().numpy(), dw0.cpu().numpy(), 2e-5) and _close(db1.cpu().numpy(), db0.cpu().numpy(), 2e-5)
        if store:
            assert _close(out.cpu().numpy(), dx.cpu().numpy(), 1e-6)
    # against fp64 autograd
    xin = torch.relu(x.double().cpu() * bnv[0].double().cpu() + bnv[1].double().cpu()) if bnv else x.double().cpu()
    w64 = torch.zeros(3, 3, Ci, Co, dtype=torch.float64, requires_grad=True)
    b64 = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    (_ref_conv(xin, w64, b64, st) * dx.double().cpu()).sum().backward()
    assert _close(dw1.cpu().numpy() - 0.5, w64.grad.numpy(), 5e-5) and _close(db1.cpu().numpy() + 0.25, b64.grad.numpy(), 5e-5)
