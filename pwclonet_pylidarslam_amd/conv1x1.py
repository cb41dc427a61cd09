"""Pointwise convolution of the module path's shared MLPs on the HIP kernels of ``csrc/conv1x1.hip`` (SURVEY.md
section 8 row f3): forward, input gradient and weight gradient on the fp32 matrix cores, on the ``(B, C, S, K)``
channel-major activations as they are (no NHWC transposes).

``conv1x1(x, weight)`` has the semantics of ``torch.nn.functional.conv{1,2,3}d(x, weight)`` for a kernel of size 1,
stride 1, no padding, no bias (P2/pytorch_utils.py:114-167 builds exactly these inside SharedMLP); anything else
stays on torch (``supported`` says which).  The three products are fp32 FMAs, so results equal torch's up to
summation order; the weight gradient is summed in a fixed order (deterministic).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


def _shape(x, weight):
    B, Cin = x.shape[0], x.shape[1]
    return B, Cin, weight.shape[0], x.numel() // max(B * Cin, 1)


def _aligned(t):
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def _fits(cin, cout):
    nbi, nbo = -(-cin // 16), -(-cout // 16)
    gy = -(-nbo // 8)
    return nbi * (-(-nbo // gy)) <= 150                      # KiB of LDS for the packed weights


def supported(x, conv):
    """True when ``conv(x)`` is a bias-free pointwise convolution these kernels cover (forward and both gradients)."""
    if not (isinstance(conv, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d)) and x.is_cuda
            and x.dtype == torch.float32 and conv.weight.dtype == torch.float32 and conv.bias is None
            and conv.groups == 1 and conv.padding_mode == "zeros"):
        return False
    if any(k != 1 for k in conv.kernel_size) or any(s != 1 for s in conv.stride) or any(d != 1 for d in conv.dilation):
        return False
    if isinstance(conv.padding, str) or any(p != 0 for p in conv.padding):
        return False
    if x.dim() != conv.weight.dim() or x.shape[1] != conv.in_channels or x.numel() == 0:
        return False
    cin, cout = conv.in_channels, conv.out_channels
    P = x.numel() // (x.shape[0] * cin)
    if P % 4 != 0 or cin > 512 or cout > 512 or x.shape[0] * (-(-P // 64)) >= 2 ** 31:
        return False
    # weight gradient: double-buffered 32-pixel chunks of all cin + cout rows in LDS, 7 float4 of staging per thread
    return (_fits(cin, cout) and _fits(cout, cin) and 2 * (cin + cout) * 36 * 4 <= 150 * 1024
            and (cin + cout) * 8 <= 7 * 512 and _wgrad_split(cin, cout))


def _wgrad_split(cin, cout):
    """csrc/conv1x1.hip wgrad_plan: the 8 waves split into wo x wm workers over the (cout/16, cin/16) tile grid, each
    owning at most 4 x 4 tiles (fewer than 5 tiles: one tile per worker, pixels split instead)."""
    nbo, nbi = -(-cout // 16), -(-cin // 16)
    if nbo * nbi < 5:
        return True
    return any(-(-nbo // wo) <= 4 and -(-nbi // (8 // wo)) <= 4 for wo in (1, 2, 4, 8))


def _forward(x, w2d, transposed, cin, cout):
    B = x.shape[0]
    P = x.numel() // (B * cin)
    y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _lib.call("conv1x1_forward_kernel_wrapper", x.device, B, cin, cout, P, x.data_ptr(), w2d.data_ptr(),
              int(transposed), y.data_ptr())
    return y


class _Conv1x1(Function):
    @staticmethod
    def forward(ctx, x, weight):
        x = _aligned(x)
        w = _aligned(weight.detach())
        B, cin, cout, P = _shape(x, w)
        ctx.save_for_backward(x, weight)
        return _forward(x, w, False, cin, cout)

    @staticmethod
    @once_differentiable            # raw kernels: a second differentiation raises instead of returning constants
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _aligned(dy)
        w = _aligned(weight.detach())
        B, cin, cout, P = _shape(x, w)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _forward(dy, w, True, cout, cin)                     # dX = W^T dY: same kernel, transposed view
        if ctx.needs_input_grad[1]:
            nbytes = _lib.load().conv1x1_wgrad_workspace_bytes(B, cin, cout, P)
            ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
            dw = torch.empty_like(w)
            _lib.call("conv1x1_wgrad_kernel_wrapper", x.device, B, cin, cout, P, dy.data_ptr(), x.data_ptr(),
                      dw.data_ptr(), ws.data_ptr())
            dw = dw.view_as(weight)
        return dx, dw


def conv1x1(x, weight):
    """``F.conv{1,2,3}d(x, weight)`` for a size-1 kernel: x (B, Cin, *), weight (Cout, Cin, 1[, 1[, 1]])."""
    return _Conv1x1.apply(x, weight)


def _folded(bn):
    """Per-channel (scale, shift) of an eval-mode BatchNorm, cached on the module until one of its tensors changes
    (version counters) or moves: a handful of tiny kernels once, not per call."""
    tensors = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors) + (bn.eps,)
    cached = getattr(bn, "_pwclo_folded", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var.float() + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight.float()
        shift = -bn.running_mean.float() * scale
        if bn.bias is not None:
            shift = shift + bn.bias.float()
        scale, shift = scale.contiguous(), shift.contiguous()
    object.__setattr__(bn, "_pwclo_folded", (key, scale, shift))
    return scale, shift


def conv1x1_bn_eval(x, conv, bn, relu):
    """Eval-mode ``act(bn(conv(x)))`` of a conv -> BatchNorm [-> ReLU] block as ONE kernel (no autograd): the running
    statistics folded to a per-channel scale / shift in the convolution's epilogue.  Same values as the three modules up
    to fp32 rounding of the folded affine map."""
    x = _aligned(x)
    w = _aligned(conv.weight.detach())
    B, cin, cout, P = _shape(x, w)
    scale, shift = _folded(bn)
    y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _lib.call("conv1x1_affine_forward_kernel_wrapper", x.device, B, cin, cout, P, x.data_ptr(), w.data_ptr(),
              scale.data_ptr(), shift.data_ptr(), int(bool(relu)), y.data_ptr())
    return y


def conv1x1_bn_eval_maxk(x, conv, bn, relu):
    """``act(bn(conv(x))).max(dim=3)[0]`` for x (B, Cin, S, K), K in {4, 8, 16, 32}, eval mode, no autograd: one
    kernel, the (B, Cout, S, K) activation is never written."""
    x = _aligned(x)
    w = _aligned(conv.weight.detach())
    B, cin, S, K = x.shape
    cout = w.shape[0]
    scale, shift = _folded(bn)
    pooled = torch.empty((B, cout, S), dtype=torch.float32, device=x.device)
    _lib.call("conv1x1_affine_maxk_forward_kernel_wrapper", x.device, B, cin, cout, S, K, x.data_ptr(), w.data_ptr(),
              scale.data_ptr(), shift.data_ptr(), int(bool(relu)), pooled.data_ptr())
    return pooled
