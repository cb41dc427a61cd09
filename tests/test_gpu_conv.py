"""Pointwise convolution kernels of the module path (csrc/conv1x1.hip) against a float64
restatement of torch.nn.functional.conv2d (what P2/pytorch_utils.py:114-167 runs) -- forward, input gradient and
weight gradient for every (cin, cout) the network's SharedMLPs hold, ragged pixel counts, determinism, and the drop-in
through the Conv2d building block."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# every bias-free Conv2d shape of PWCLO-Net (tests/test_host_cpu.py pins this list against the model)
NET_SHAPES = [(6, 8), (8, 8), (8, 16), (10, 64), (16, 16), (16, 32), (19, 16), (32, 32), (32, 64), (35, 32), (42, 128),
              (64, 64), (64, 128), (67, 64), (67, 128), (74, 128), (80, 64), (96, 64), (128, 64), (128, 128),
              (138, 128), (144, 128), (160, 128), (192, 128)]


def _ref(x, w, dy):
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    y = torch.einsum("oi,bip->bop", w64, x64)
    dx = torch.einsum("oi,bop->bip", w64, dy64)
    dw = torch.einsum("bop,bip->oi", dy64, x64)
    return y, dx, dw


def _run(x, w, dy):
    from pwclonet_pylidarslam_amd import conv1x1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = conv1x1.conv1x1(xr, wr.view(w.shape[0], w.shape[1], 1))
    y.backward(dy)
    return y.detach(), xr.grad, wr.grad


def _check(got, ref, what, rel=1e-5):
    # fp32 FMAs in another order than a float64 sum: bound relative to the tensor's scale
    scale = ref.abs().max().item()
    err = (got.double() - ref).abs().max().item()
    assert err <= rel * scale + 1e-30, (what, err, scale)
    return err / max(scale, 1e-30)


@pytest.mark.parametrize("cin,cout", NET_SHAPES)
def test_conv1x1_forward_and_gradients_match_float64(cuda, cin, cout):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    B, P = 3, 2052                                           # 2052 = 32 tiles of 64 + a ragged tail of 4 pixels
    x = torch.randn(B, cin, P, generator=g).to(cuda)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(cuda)
    dy = torch.randn(B, cout, P, generator=g).to(cuda)
    y, dx, dw = _run(x, w, dy)
    ry, rdx, rdw = _ref(x, w, dy)
    _check(y, ry, "y")
    _check(dx, rdx, "dx")
    _check(dw, rdw, "dw")


@pytest.mark.parametrize("B,P", [(1, 4), (2, 8), (1, 60), (5, 64), (2, 68), (1, 16384), (7, 1028)])
def test_conv1x1_ragged_pixel_counts(cuda, B, P):
    g = torch.Generator().manual_seed(B * 100000 + P)
    for cin, cout in ((6, 8), (67, 128), (192, 128)):
        x = torch.randn(B, cin, P, generator=g).to(cuda)
        w = torch.randn(cout, cin, generator=g).to(cuda)
        dy = torch.randn(B, cout, P, generator=g).to(cuda)
        y, dx, dw = _run(x, w, dy)
        ry, rdx, rdw = _ref(x, w, dy)
        _check(y, ry, "y")
        _check(dx, rdx, "dx")
        _check(dw, rdw, "dw")


@pytest.mark.parametrize("cin,cout", [(1, 1), (3, 4), (2, 5)])
@pytest.mark.parametrize("P", [4, 64, 128])
def test_conv1x1_few_channels_few_pixels(cuda, cin, cout, P):
    """with fewer than 8 rows (cin + cout) and <= 128 pixels the weight gradient's phase reduction reused a
    staging area smaller than its 8 KiB of phase sums (stores beyond the reserved LDS were dropped: dW came back wrong,
    silently).  wgrad_plan now reserves max(staging, phase sums); these are the shapes that reached it."""
    g = torch.Generator().manual_seed(1000 * cin + 10 * cout + P)
    x = torch.randn(2, cin, P, generator=g).to(cuda)
    w = torch.randn(cout, cin, generator=g).to(cuda)
    dy = torch.randn(2, cout, P, generator=g).to(cuda)
    y, dx, dw = _run(x, w, dy)
    ry, rdx, rdw = _ref(x, w, dy)
    _check(y, ry, "y")
    _check(dx, rdx, "dx")
    _check(dw, rdw, "dw")


def test_conv1x1_large_layer_is_deterministic_and_close_to_torch(cuda):
    """A full-size layer of the level-1 set-upconv (B=8 here): 128 -> 128 over 131072 pixels.  Two runs are
    bit-identical (fixed summation order, no atomics); against torch's own convolution the three results agree to the
    fp32 summation-order bound; timing of both printed for information."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    B, cin, cout, S, K = 8, 128, 128, 2048, 8
    x = torch.randn(B, cin, S, K, generator=g).to(cuda)
    w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(cuda)
    dy = torch.randn(B, cout, S, K, generator=g).to(cuda)
    from pwclonet_pylidarslam_amd import conv1x1

    def ours():
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = conv1x1.conv1x1(xr, wr)
        y.backward(dy)
        return y.detach(), xr.grad, wr.grad

    def stock():
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = F.conv2d(xr, wr)
        y.backward(dy)
        return y.detach(), xr.grad, wr.grad
    a, b, t = ours(), ours(), stock()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for u, v, name in zip(a, t, ("y", "dx", "dw")):
        scale = v.abs().max().item()
        assert (u - v).abs().max().item() <= 2e-5 * scale, name
    times = {}
    for name, fn in (("hip", ours), ("torch", stock)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / 5
    print("\nconv 128->128 over 8x2048x8 pixels, forward + both gradients (+2 clones): hip %.3f ms, torch %.3f ms"
          % (times["hip"], times["torch"]))


def test_conv_block_routes_recorded_layers_through_the_kernels(cuda):
    """pytorch_utils.Conv2d (P2/pytorch_utils.py:170-199): with autograd recording, the bias-free 1x1 convolution
    runs on csrc/conv1x1.hip and gives torch's gradients; a layer with a bias stays on torch."""
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(3)
    blk = pt.Conv2d(19, 16, bn=True).to(cuda).train()
    x = torch.randn(2, 19, 64, 8, device=cuda, requires_grad=True)
    seen = []
    orig = pt._hip_conv.conv1x1
    pt._hip_conv.conv1x1 = lambda *a: (seen.append(1), orig(*a))[1]
    try:
        y = blk(x)
        y.square().sum().backward()
        gx, gw = x.grad.clone(), blk.conv.weight.grad.clone()
        assert seen == [1]
        withbias = pt.Conv2d(19, 16, bn=False).to(cuda)
        withbias(x)
        assert seen == [1]
    finally:
        pt._hip_conv.conv1x1 = orig
    old = pt._USE_HIP_CONV
    pt._USE_HIP_CONV = "0"
    try:
        blk.zero_grad()
        x.grad = None
        blk2 = blk                                            # same parameters, torch convolution
        y2 = blk2(x)
        y2.square().sum().backward()
    finally:
        pt._USE_HIP_CONV = old
    assert (y - y2).abs().max().item() <= 1e-5 * y2.abs().max().item()
    assert (gx - x.grad).abs().max().item() <= 1e-4 * x.grad.abs().max().item()
    assert (gw - blk.conv.weight.grad).abs().max().item() <= 1e-4 * blk.conv.weight.grad.abs().max().item()


@pytest.mark.parametrize("B,C,S,K", [(2, 16, 64, 32), (3, 5, 36, 16), (1, 64, 256, 8), (2, 7, 12, 4), (4, 32, 1024, 32)])
def test_bn_relu_max_tail_matches_the_three_separate_ops(cuda, B, C, S, K):
    """csrc/batchnorm.hip tail (BatchNorm with batch statistics -> ReLU -> max over K) against torch's own three ops in
    float64: pooled values, running statistics, and the gradients w.r.t. the input, gamma and beta.  Bound 1e-5 of each
    tensor's scale (fp32 arithmetic with fp64 statistics against a float64 evaluation)."""
    import torch.nn.functional as F
    from pwclonet_pylidarslam_amd import batchnorm as hb
    g = torch.Generator().manual_seed(B * 1000 + C * 10 + K)
    x = (torch.randn(B, C, S, K, generator=g) * 1.7 + 0.3).to(cuda)
    bn = torch.nn.BatchNorm2d(C).to(cuda).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    dpool = torch.randn(B, C, S, generator=g).to(cuda)
    ref_bn = torch.nn.BatchNorm2d(C).to(cuda).double().train()
    ref_bn.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    xr = x.double().requires_grad_(True)
    pr = F.relu(ref_bn(xr)).max(dim=3)[0]
    pr.backward(dpool.double())
    xh = x.clone().requires_grad_(True)
    assert hb.supported_maxk(xh, bn)
    ph = hb.batch_norm_train_relu_max(xh, bn)
    ph.backward(dpool)

    def close(a, r, what):
        scale = r.abs().max().item()
        err = (a.double() - r).abs().max().item()
        assert err <= 1e-5 * scale + 1e-12, (what, err, scale)
    close(ph, pr.detach(), "pooled")
    close(bn.running_mean, ref_bn.running_mean, "running_mean")
    close(bn.running_var, ref_bn.running_var, "running_var")
    assert int(bn.num_batches_tracked) == 1
    close(xh.grad, xr.grad, "dx")
    close(bn.weight.grad, ref_bn.weight.grad, "dgamma")
    close(bn.bias.grad, ref_bn.bias.grad, "dbeta")


def test_shared_mlp_max_is_the_stack_followed_by_max(cuda):
    """pytorch_utils.shared_mlp_max on a training-mode SharedMLP equals mlp(x).max(dim=3)[0] -- values, running
    statistics and parameter gradients -- and falls back to exactly that in eval mode."""
    import copy
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(11)
    mlp = pt.SharedMLP([19, 16, 16, 32], bn=True).to(cuda).train()
    ref = copy.deepcopy(mlp)
    x = torch.randn(2, 19, 128, 16, device=cuda)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = pt.shared_mlp_max(mlp, xa)
    saved = (pt._USE_HIP_STACK, pt._USE_HIP_CONV, pt._USE_HIP_BN)
    pt._USE_HIP_STACK, pt._USE_HIP_CONV, pt._USE_HIP_BN = False, "0", False     # the reference side: torch's own ops
    try:
        b = ref(xb).max(dim=3)[0]
    finally:
        pt._USE_HIP_STACK, pt._USE_HIP_CONV, pt._USE_HIP_BN = saved
    assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
    w = torch.randn_like(a)
    (a * w).sum().backward()
    (b * w).sum().backward()
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-4 * xb.grad.abs().max().item()
    for (n, p), (_, q) in zip(mlp.named_parameters(), ref.named_parameters()):
        assert (p.grad - q.grad).abs().max().item() <= 1e-4 * q.grad.abs().max().item() + 1e-7, n
    for (n, p), (_, q) in zip(mlp.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-5, atol=1e-6), n
    mlp.eval()
    ref.eval()
    with torch.no_grad():
        assert torch.equal(pt.shared_mlp_max(mlp, x), mlp(x).max(dim=3)[0])


@pytest.mark.parametrize("na,nb", [(1, 257), (257, 1), (64, 64), (1, 1)])
def test_hamilton_product_op_is_the_reference_expression(cuda, na, nb):
    """PWCLO_utils._hamilton on csrc/warp.hip hamilton_kernel: forward bit-identical to the reference's component
    expressions (PW/PWCLO_utils.py:83-95), first- and second-order gradients equal to autograd's on the expression
    (broadcast operands included)."""
    from pwclonet_pylidarslam_amd.pwclonet import PWCLO_utils as U

    def expr(a, b):
        a0, a1, a2, a3 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        b0, b1, b2, b3 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        return torch.stack((a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                            a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1, a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0), dim=1)
    g = torch.Generator().manual_seed(na * 7 + nb)
    a = torch.randn(3, 4, na, generator=g).to(cuda)
    b = torch.randn(3, 4, nb, generator=g).to(cuda)
    w = torch.randn(3, 4, max(na, nb), generator=g).to(cuda)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    c1, c2 = U._hamilton(a1, b1), expr(a2, b2)
    assert torch.equal(c1, c2)
    ga1, gb1 = torch.autograd.grad((c1 * w).sum(), (a1, b1), create_graph=True)
    ga2, gb2 = torch.autograd.grad((c2 * w).sum(), (a2, b2), create_graph=True)
    for u, v in ((ga1, ga2), (gb1, gb2)):
        assert (u - v).abs().max().item() <= 1e-5 * v.abs().max().item()
    # second order: d/da of sum(ga * gb-ish) exercises the conjugate flags
    s1 = (ga1.square().sum() + gb1.square().sum())
    s2 = (ga2.square().sum() + gb2.square().sum())
    h1 = torch.autograd.grad(s1, (a1, b1))
    h2 = torch.autograd.grad(s2, (a2, b2))
    for u, v in zip(h1, h2):
        assert (u - v).abs().max().item() <= 1e-4 * v.abs().max().item()


def test_eval_block_conv_bn_relu_is_one_kernel_with_the_modules_values(cuda):
    """Eval mode, nothing recorded: pytorch_utils.Conv2d(bn=True) runs convolution + folded BatchNorm + ReLU as one
    kernel (conv1x1_affine_forward); values equal the three torch modules' within 1e-5 of the output scale, the folded
    parameters follow an in-place edit of the running statistics, and a recorded call still takes the autograd route."""
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(5)
    blk = pt.Conv2d(67, 128, bn=True).to(cuda)
    bn = blk.bn.bn
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    blk.eval()
    x = torch.randn(3, 67, 100, 8, device=cuda)

    def stock():
        with torch.no_grad():
            return torch.relu(bn(torch.nn.functional.conv2d(x, blk.conv.weight)))
    with torch.no_grad():
        y = blk(x)
    ref = stock()
    assert (y - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    with torch.no_grad():
        bn.running_mean.add_(0.25)                           # in-place edit: the cached fold must follow
        y2 = blk(x)
    ref2 = stock()
    assert (y2 - ref2).abs().max().item() <= 1e-5 * ref2.abs().max().item()
    assert (ref2 - ref).abs().max().item() > 1e-3
    xr = x.clone().requires_grad_(True)
    yr = blk(xr)                                             # recorded: separate ops, differentiable
    assert yr.requires_grad and (yr - ref2).abs().max().item() <= 1e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("K,S", [(32, 100), (16, 36), (8, 257), (4, 64)])
def test_eval_stack_tail_conv_bn_relu_max_in_one_kernel(cuda, K, S):
    """pytorch_utils.shared_mlp_max in eval mode with nothing recorded: the last layer's convolution, folded BatchNorm,
    ReLU and the max over K as one kernel (conv1x1_affine_maxk_forward) -- equal to the modules followed by
    .max(dim=3)[0] within 1e-5 of the output scale; ragged S (tiles that end inside a batch element) included."""
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(K + S)
    mlp = pt.SharedMLP([35, 32, 32, 64], bn=True).to(cuda)
    for m in mlp.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_()
    mlp.eval()
    x = torch.randn(3, 35, S, K, device=cuda)
    old = pt._USE_HIP_CONV
    try:
        pt._USE_HIP_CONV = "0"
        with torch.no_grad():
            ref = mlp(x).max(dim=3)[0]
        pt._USE_HIP_CONV = "all"
        with torch.no_grad():
            got = pt.shared_mlp_max(mlp, x)
    finally:
        pt._USE_HIP_CONV = old
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_conv1x1_supported_matches_what_the_launchers_accept(cuda):
    """conv1x1.supported() (Python) restates the launchers' limits (LDS for the packed weights, tiles per wave of the
    weight gradient): for random channel counts up to 400 every shape it accepts runs forward and both gradients without
    a library error and matches float64; a shape it rejects is left to torch by the Conv blocks."""
    from pwclonet_pylidarslam_amd import _lib
    from pwclonet_pylidarslam_amd import conv1x1
    g = torch.Generator().manual_seed(77)
    accepted = 0
    for _ in range(24):
        cin = int(torch.randint(1, 401, (1,), generator=g))
        cout = int(torch.randint(1, 401, (1,), generator=g))
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(cuda)
        x = torch.randn(2, cin, 16, 8, generator=g).to(cuda)
        if not conv1x1.supported(x, conv):
            continue
        accepted += 1
        _lib.synchronize(cuda)
        dy = torch.randn(2, cout, 128, generator=g).to(cuda)
        y, dx, dw = _run(x.flatten(2), conv.weight.detach().flatten(1), dy)
        ry, rdx, rdw = _ref(x.flatten(2), conv.weight.detach().flatten(1), dy)
        _check(y, ry, "y %d->%d" % (cin, cout))
        _check(dx, rdx, "dx %d->%d" % (cin, cout))
        _check(dw, rdw, "dw %d->%d" % (cin, cout))
    _lib.synchronize(cuda)
    assert accepted >= 6
    big = torch.nn.Conv2d(600, 64, 1, bias=False).to(cuda)
    assert not conv1x1.supported(torch.randn(1, 600, 4, 4, device=cuda), big)
    # 32-bit byte offsets into x / dy: 4 GiB and beyond stays on torch (shape check only: no such tensor is allocated here)
    wide = torch.nn.Conv2d(16, 16, 1, bias=False)
    assert conv1x1.supported_layer(wide, 64, 65536 * 16 - 64, 4)
    assert not conv1x1.supported_layer(wide, 64, 65536 * 16, 4)


@pytest.mark.parametrize("pooled,K", [(True, 16), (False, 8), (True, 32), (False, 4)])
def test_training_stack_without_normalised_activations_matches_torch(cuda, pooled, K):
    """pytorch_utils._train_stack: a SharedMLP in training mode as conv -> [BN+ReLU fused into the next conv] ... ->
    BN+ReLU(+max) with no normalised activation written, against the same stack evaluated by torch's own conv2d /
    batch_norm / relu / max in float64: output, running statistics, input gradient and every parameter gradient.
    Bounds: 1e-5 of the scale for values, 1e-4 for gradients (three BatchNorm backward passes in fp32)."""
    import copy
    import torch.nn.functional as F
    from pwclonet_pylidarslam_amd.pointnet2_ops import pytorch_utils as pt
    torch.manual_seed(3 + K)
    mlp = pt.SharedMLP([35, 32, 48, 64], bn=True).to(cuda).train()
    for m in mlp.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.3)
    ref = copy.deepcopy(mlp).double()
    x = torch.randn(3, 35, 40, K, device=cuda)
    xa = x.clone().requires_grad_(True)
    out = pt._train_stack(mlp, xa, pooled)
    assert out is not None
    xr = x.double().requires_grad_(True)
    h = xr
    for layer in ref:
        h = F.conv2d(h, layer.conv.weight)
        bn = layer.bn.bn
        h = F.relu(F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps))
    want = h.max(dim=3)[0] if pooled else h
    assert out.shape == want.shape
    assert (out.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    w = torch.randn_like(out)
    (out * w).sum().backward()
    (want * w.double()).sum().backward()
    assert (xa.grad.double() - xr.grad).abs().max().item() <= 1e-4 * xr.grad.abs().max().item()
    for (n, p), (_, q) in zip(mlp.named_parameters(), ref.named_parameters()):
        assert (p.grad.double() - q.grad).abs().max().item() <= 1e-4 * q.grad.abs().max().item() + 1e-9, n
    for (n, p), (_, q) in zip(mlp.named_buffers(), ref.named_buffers()):
        if p.is_floating_point():
            assert (p.double() - q).abs().max().item() <= 1e-5 * max(q.abs().max().item(), 1.0), n
        else:
            assert int(p) == 1, n                             # num_batches_tracked (the float64 copy was not advanced)


@pytest.mark.parametrize("B,cin,cout,P", [(2, 16, 8, 4096), (3, 48, 35, 1028), (2, 64, 128, 512), (1, 5, 3, 4), (2, 35, 64, 260),
                                          (32, 16, 8, 65536)])
@pytest.mark.parametrize("affine", [True, False])
def test_input_gradient_epilogue_leaves_the_batchnorm_backward_sums(cuda, B, cin, cout, P, affine):
    """conv1x1_dgrad_bnstats + batchnorm_train_backward_apply against the pair they replace (input-gradient convolution,
    then batchnorm_train_backward with its own reduction pass): da bit-identical, dgamma / dbeta within 1e-5 of their
    scale (per-lane fp32 partial sums of <= 64 values instead of fp64 throughout), dx within 1e-5 of its scale."""
    from pwclonet_pylidarslam_amd import _lib, conv1x1
    from pwclonet_pylidarslam_amd import batchnorm as hb
    gen = torch.Generator().manual_seed(B * 13 + cin + cout)
    x = torch.randn(B, cin, P, generator=gen).to(cuda) * 2 + 0.5            # the BatchNorm's input
    dy = torch.randn(B, cout, P, generator=gen).to(cuda)
    w = (torch.randn(cout, cin, generator=gen) / cin ** 0.5).to(cuda)
    mean = x.mean(dim=(0, 2)).contiguous()
    invstd = (1.0 / torch.sqrt(x.var(dim=(0, 2), unbiased=False) + 1e-5)).contiguous()
    gamma = (torch.rand(cin, generator=gen) + 0.5).to(cuda) if affine else None
    beta = (torch.randn(cin, generator=gen) * 0.3).to(cuda) if affine else None
    p = lambda t: t.data_ptr() if t is not None else 0
    da_ref = conv1x1._forward(dy, w, True, cout, cin)
    dx_ref, dg_ref, db_ref = torch.empty_like(x), torch.empty(cin, device=cuda), torch.empty(cin, device=cuda)
    _lib.call("batchnorm_train_backward_kernel_wrapper", cuda, B, cin, P, p(x), p(da_ref), p(gamma), p(beta), p(mean), p(invstd),
              p(dx_ref), p(dg_ref), p(db_ref), p(hb._workspace(cin, cuda)), 1)
    da, dx, dg, db = torch.empty_like(x), torch.empty_like(x), torch.empty(cin, device=cuda), torch.empty(cin, device=cuda)
    ws = torch.empty((_lib.load().conv1x1_stats_workspace_bytes(B, cout, cin, P) // 8,), dtype=torch.float64, device=cuda)
    _lib.call("conv1x1_dgrad_bnstats_kernel_wrapper", cuda, B, cin, cout, P, p(dy), p(w), p(x), p(mean), p(invstd), p(gamma),
              p(beta), p(da), p(dg), p(db), p(ws))
    _lib.call("batchnorm_train_backward_apply_kernel_wrapper", cuda, B, cin, P, p(x), p(da), p(gamma), p(beta), p(mean), p(invstd),
              p(dg), p(db), p(dx), 1)
    assert torch.equal(da, da_ref)
    for got, ref, what in ((dg, dg_ref, "dgamma"), (db, db_ref, "dbeta"), (dx, dx_ref, "dx")):
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 1e-5 * scale + 1e-30, (what, (got - ref).abs().max().item(), scale)


@pytest.mark.parametrize("cin,cout,P", [(19, 16, 64), (67, 128, 36), (3, 5, 4), (138, 128, 1028)])
def test_conv1x1_channel_tail_does_not_read_the_next_cloud(cuda, cin, cout, P):
    """The kernel reads its input in 16-channel blocks through a buffer descriptor; the lanes of the last block that
    stand for channels >= cin would alias the NEXT cloud's first channels.  They must contribute exact zeros: with NaNs in
    exactly those aliased rows (cloud 1, channels 0 .. 15 - cin % 16) cloud 0's output stays finite and equal to the
    float64 product, forward and input gradient (whose "channels" are the layer's cout) alike; cloud 1's is NaN."""
    from pwclonet_pylidarslam_amd import conv1x1
    gen = torch.Generator().manual_seed(cin * 7 + P)
    x = torch.randn(2, cin, P, generator=gen)
    w = torch.randn(cout, cin, generator=gen) / cin ** 0.5
    alias = 16 - cin % 16 if cin % 16 else 0
    x[1, :max(alias, 1)] = float("nan")
    xc, wc = x.to(cuda), w.to(cuda)
    y = conv1x1._forward(xc, wc, False, cin, cout)
    want = torch.einsum("oi,ip->op", w.double(), x[0].double())
    assert torch.isfinite(y[0]).all()
    assert (y[0].cpu().double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert torch.isnan(y[1]).all()
    dy = torch.randn(2, cout, P, generator=gen)
    alias_o = 16 - cout % 16 if cout % 16 else 0
    dy[1, :max(alias_o, 1)] = float("nan")
    dx = conv1x1._forward(dy.to(cuda), wc, True, cout, cin)
    want_dx = torch.einsum("oi,op->ip", w.double(), dy[0].double())
    assert torch.isfinite(dx[0]).all()
    assert (dx[0].cpu().double() - want_dx).abs().max().item() <= 1e-5 * want_dx.abs().max().item()
    assert torch.isnan(dx[1]).all()


@pytest.mark.parametrize("B,cin,cout,P,shift", [(2, 6, 8, 4096, 0.0), (3, 35, 48, 1028, 0.0), (1, 64, 128, 64, 0.0),
                                                 (2, 138, 128, 512, 0.0), (2, 19, 200, 260, 0.0), (1, 3, 5, 4, 0.0),
                                                 (64, 8, 16, 65536, 3.0), (4, 67, 64, 16384, 5.0)])
@pytest.mark.parametrize("transform", [False, True])
def test_conv_epilogue_batch_statistics(cuda, B, cin, cout, P, shift, transform):
    """conv1x1_forward_bnstats: the convolution's output is bit-identical to the plain kernel's, and the batch statistics
    summed in its epilogue (per-lane fp32 over <= 64 values, fp64 beyond) equal a float64 evaluation of
    F.batch_norm(training=True)'s mean / biased variance / running update on that output: 2e-6 of |mean| + std, 5e-6
    relative for 1/sqrt(var + eps) -- also where |mean| is several standard deviations (``shift``) and on the largest
    activation of the training step (64 x 16 x 65536: 32 tiles per wave)."""
    from pwclonet_pylidarslam_amd import _lib, conv1x1
    if B * (cin + cout) * P * 4 > 2 ** 31:
        pytest.skip("too large")
    gen = torch.Generator().manual_seed(cin * 131 + cout)
    x = (torch.randn(B, cin, P, generator=gen) + shift).to(cuda)
    w = (torch.randn(cout, cin, generator=gen) / cin ** 0.5).to(cuda)
    tf = None
    if transform:
        tf = tuple(t.to(cuda) for t in (torch.randn(cin, generator=gen), torch.rand(cin, generator=gen) + 0.5,
                                        torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen) * 0.3))
    rm = torch.randn(cout, generator=gen).to(cuda)
    rv = (torch.rand(cout, generator=gen) + 0.5).to(cuda)
    rm0, rv0 = rm.clone(), rv.clone()
    eps, mom = 1e-5, 0.1
    y, mean, invstd = conv1x1._forward_stats(x, w, cin, cout, tf, rm, rv, mom, eps)
    if transform:
        plain = torch.empty_like(y)
        _lib.call("conv1x1_bnrelu_forward_kernel_wrapper", x.device, B, cin, cout, P, x.data_ptr(), w.data_ptr(),
                  tf[0].data_ptr(), tf[1].data_ptr(), tf[2].data_ptr(), tf[3].data_ptr(), plain.data_ptr())
    else:
        plain = conv1x1._forward(x, w, False, cin, cout)
    assert torch.equal(y, plain)
    y64 = y.double()
    m64 = y64.mean(dim=(0, 2))
    v64 = y64.var(dim=(0, 2), unbiased=False)
    sd = v64.sqrt()
    assert ((mean.double() - m64).abs() <= 2e-6 * (m64.abs() + sd) + 1e-12).all(), ((mean.double() - m64).abs() / (m64.abs() + sd)).max()
    is64 = 1.0 / torch.sqrt(v64 + eps)
    rel = ((invstd.double() - is64).abs() / is64).max().item()
    assert rel <= 5e-6, rel
    n = B * P
    unb = v64 * n / max(n - 1, 1)
    torch.testing.assert_close(rm.double(), (1 - mom) * rm0.double() + mom * m64, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv.double(), (1 - mom) * rv0.double() + mom * unb, rtol=1e-5, atol=1e-6)
    y2, mean2, invstd2 = conv1x1._forward_stats(x, w, cin, cout, tf, None, None, mom, eps)      # deterministic
    assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2)


@pytest.mark.parametrize("K", [4, 6, 32, 8])
def test_softmax_weighted_sum_matches_torch(cuda, K):
    """csrc/softmax_wsum.hip behind pwclonet/costvolume.py: sum(softmax(x, dim=3) * v, dim=3), forward and both gradients,
    against the float64 evaluation of the three torch ops (PW/costvolume.py:139-141, 181-183); ragged row counts; a K the
    kernel is not built for falls back to torch."""
    from pwclonet_pylidarslam_amd.softmax_wsum import softmax_weighted_sum
    g = torch.Generator().manual_seed(40 + K)
    for shape in ((2, 64, 301, K), (1, 3, 7, K)):
        x = (torch.randn(shape, generator=g) * 3).to(cuda).requires_grad_(True)
        v = torch.randn(shape, generator=g).to(cuda).requires_grad_(True)
        go = torch.randn(shape[:3], generator=g).to(cuda)
        out = softmax_weighted_sum(x, v)
        out.backward(go)
        x64, v64 = x.detach().double().requires_grad_(True), v.detach().double().requires_grad_(True)
        ref = torch.sum(torch.softmax(x64, dim=3) * v64, dim=3)
        ref.backward(go.double())
        _check(out.detach(), ref.detach(), "out", rel=2e-6)
        _check(x.grad, x64.grad, "dx", rel=2e-6)
        _check(v.grad, v64.grad, "dv", rel=2e-6)
    x = torch.randn(2, 5, 9, 5, generator=g).to(cuda)            # K = 5: torch ops
    v = torch.randn(2, 5, 9, 5, generator=g).to(cuda)
    torch.testing.assert_close(softmax_weighted_sum(x, v), torch.sum(torch.softmax(x, dim=3) * v, dim=3))
