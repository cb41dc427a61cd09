"""Training-mode BatchNorm on the HIP kernels of ``csrc/batchnorm.hip``.

``batch_norm_train`` has the semantics of ``torch.nn.functional.batch_norm(..., training=True)`` for float32
``(B, C, *)`` CUDA/HIP tensors and is what the ``BatchNorm1d/2d`` wrappers of ``pointnet2_ops.pytorch_utils`` call
in training mode: the module path's activations are ``(B, C, S, K)`` with few channels and very long rows, a shape
the stock kernels run an order of magnitude below the HBM rate.  Evaluation mode is untouched (plain torch, and
folded into the packed weights on the fused path).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


# ``num_batches_tracked += 1`` is one tiny launch per BatchNorm call -- 93 per PWCLO-Net training forward (0.4 ms of kernel
# time and as many host launches).  Inside ``deferred_counters()`` the calls are only recorded and applied as ONE
# multi-tensor add when the block ends (a layer called twice in the block gets += 2).  The counters are not read by these
# kernels (they require ``momentum is not None``), so nothing inside the block can observe the delay.
_pending = None


class deferred_counters:
    def __enter__(self):
        global _pending
        self._outer = _pending
        _pending = {} if _pending is None else _pending
        return self

    def __exit__(self, *exc):
        global _pending
        pend, _pending = _pending, self._outer
        if self._outer is None and pend:
            tensors = [t for t, _ in pend.values()]
            torch._foreach_add_(tensors, [int(c) for _, c in pend.values()])
        return False


def count_batch(bn):
    """``bn.num_batches_tracked += 1`` (now, or at the end of the enclosing ``deferred_counters()`` block)."""
    if not (bn.track_running_stats and bn.num_batches_tracked is not None):
        return
    if _pending is None:
        bn.num_batches_tracked.add_(1)
    else:
        t = bn.num_batches_tracked
        ent = _pending.get(id(t))
        _pending[id(t)] = (t, (ent[1] if ent else 0) + 1)


def _touch(*tensors):
    """The kernels update running_mean / running_var through raw pointers: bump their version counters so that
    anything keyed on them (conv1x1._folded) sees the write."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _workspace(c, device):
    nbytes = _lib.load().batchnorm_train_workspace_bytes(int(c))
    return torch.empty((nbytes // 8,), dtype=torch.float64, device=device)


class _BatchNormTrain(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, given_mean=None, given_invstd=None):
        if not x.is_cuda:
            raise RuntimeError("CPU not supported")
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        y = torch.empty_like(x)
        p = lambda t: t.data_ptr() if t is not None else 0
        if given_mean is not None:       # the producing convolution's epilogue already has the statistics (conv1x1_stats)
            save_mean, save_invstd = given_mean, given_invstd
            _lib.call("batchnorm_train_apply_kernel_wrapper", x.device, B, C, L, p(x), p(weight), p(bias), p(save_mean),
                      p(save_invstd), p(y), int(relu))
        else:
            save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
            save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
            ws = _workspace(C, x.device)
            _lib.call("batchnorm_train_forward_kernel_wrapper", x.device, B, C, L, p(x), p(weight), p(bias), float(eps),
                      float(momentum), p(running_mean), p(running_var), p(y), p(save_mean), p(save_invstd), p(ws), int(relu))
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    @once_differentiable            # raw kernels: a second differentiation raises instead of returning constants
    def backward(ctx, dy):
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        dy = dy.contiguous()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = _workspace(C, x.device)
        p = lambda t: t.data_ptr() if t is not None else 0
        _lib.call("batchnorm_train_backward_kernel_wrapper", x.device, B, C, L, p(x), p(dy), p(weight), p(bias),
                  p(save_mean), p(save_invstd), p(dx), p(dgamma), p(dbeta), p(ws), int(ctx.relu))
        return (dx, (dgamma if weight is not None else None), (dbeta if weight is not None else None), None, None, None,
                None, None, None, None)


def supported(x, bn):
    """The custom path covers what the PWCLO-Net stacks use; anything else stays on torch."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and x.numel() > 0 and bn.momentum is not None
            and (bn.weight is None) == (bn.bias is None)
            and (not bn.track_running_stats or bn.running_mean is not None))


def batch_norm_train(x, bn, relu=False, stats=None):
    """Training-mode forward of the ``torch.nn.BatchNorm*`` module ``bn`` on ``x`` (updates its running statistics
    and ``num_batches_tracked`` like ``bn(x)`` does); ``relu=True`` also applies the stack's ReLU in the same pass.
    ``stats = (mean, invstd)``: the batch statistics of ``x`` as ``conv1x1.conv1x1_stats`` returned them (that call has
    already updated the running statistics and the counter): only the apply pass runs."""
    if stats is not None:
        return _BatchNormTrain.apply(x, bn.weight, bn.bias, None, None, bn.momentum, bn.eps, relu, stats[0], stats[1])
    count_batch(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    y = _BatchNormTrain.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, relu)
    _touch(rm, rv)
    return y


class _BatchNormReluMaxK(Function):
    """BatchNorm (batch statistics) -> ReLU -> max over the last dimension of x (B, C, S, K), the tail of the grouped
    stacks (P2/pointnet2_modules.py: SharedMLP followed by ``.max(dim=3)``), without the (B, C, S, K) activation:
    the forward keeps the arg-max and the selected inputs, the backward rebuilds the sparse gradient from them."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, given_mean=None, given_invstd=None):
        x = x.contiguous()
        B, C, S, K = x.shape
        pooled = torch.empty((B, C, S), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C, S), dtype=torch.uint8, device=x.device)
        xsel = torch.empty((B, C, S), dtype=torch.float32, device=x.device)
        p = lambda t: t.data_ptr() if t is not None else 0
        if given_mean is not None:
            save_mean, save_invstd = given_mean, given_invstd
            _lib.call("batchnorm_train_relu_maxk_apply_kernel_wrapper", x.device, B, C, S, K, p(x), p(weight), p(bias),
                      p(save_mean), p(save_invstd), p(pooled), p(arg), p(xsel))
        else:
            save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
            save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
            ws = _workspace(C, x.device)
            _lib.call("batchnorm_train_relu_maxk_forward_kernel_wrapper", x.device, B, C, S, K, p(x), p(weight), p(bias),
                      float(eps), float(momentum), p(running_mean), p(running_var), p(pooled), p(arg), p(xsel),
                      p(save_mean), p(save_invstd), p(ws))
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd, arg, xsel)
        return pooled

    @staticmethod
    @once_differentiable            # raw kernels: a second differentiation raises instead of returning constants
    def backward(ctx, dpool):
        x, weight, bias, save_mean, save_invstd, arg, xsel = ctx.saved_tensors
        dpool = dpool.contiguous()
        B, C, S, K = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = _workspace(C, x.device)
        p = lambda t: t.data_ptr() if t is not None else 0
        _lib.call("batchnorm_train_relu_maxk_backward_kernel_wrapper", x.device, B, C, S, K, p(x), p(dpool), p(arg),
                  p(xsel), p(weight), p(bias), p(save_mean), p(save_invstd), p(dx), p(dgamma), p(dbeta), p(ws))
        return (dx, (dgamma if weight is not None else None), (dbeta if weight is not None else None), None, None, None,
                None, None, None)


def supported_maxk(x, bn):
    return supported(x, bn) and x.dim() == 4 and x.shape[3] in (4, 8, 16, 32)


def batch_norm_train_relu_max(x, bn, stats=None):
    """``relu(bn(x)).max(dim=3)[0]`` for the training-mode module ``bn`` and x (B, C, S, K): one statistics pass and one
    pooled pass, nothing of shape (B, C, S, K) written.  ``stats``: see ``batch_norm_train``."""
    if stats is not None:
        return _BatchNormReluMaxK.apply(x, bn.weight, bn.bias, None, None, bn.momentum, bn.eps, stats[0], stats[1])
    count_batch(bn)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    y = _BatchNormReluMaxK.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps)
    _touch(rm, rv)
    return y
