"""``sum(softmax(logits, dim=3) * values, dim=3)`` as one kernel each way (csrc/softmax_wsum.hip) -- the tail of both
aggregates of the attentive cost volume (PW/costvolume.py:139-141, 181-183) on the module path.

``softmax_weighted_sum(logits, values)`` has the value and the gradients of the three torch ops it replaces (softmax,
multiply, sum) up to fp32 rounding; anything the kernel does not cover (CPU tensors, other dtypes, K it is not built for)
runs those torch ops.  Nothing but the two inputs is saved for backward: the probabilities are recomputed.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib


def supported(logits, values):
    return (logits.is_cuda and values.is_cuda and logits.dtype == torch.float32 and values.dtype == torch.float32
            and logits.dim() == 4 and logits.shape == values.shape and logits.numel() > 0
            and bool(_lib.load().softmax_wsum_supported_k(int(logits.shape[3]))))


def _aligned(t):
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


class _SoftmaxWeightedSum(Function):
    @staticmethod
    def forward(ctx, logits, values):
        x, v = _aligned(logits), _aligned(values)
        B, C, S, K = x.shape
        out = torch.empty((B, C, S), dtype=torch.float32, device=x.device)
        _lib.call("softmax_wsum_forward_kernel_wrapper", x.device, B * C * S, K, x.data_ptr(), v.data_ptr(), out.data_ptr())
        ctx.save_for_backward(x, v)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, v = ctx.saved_tensors
        B, C, S, K = x.shape
        dout = dout.contiguous()
        dx, dv = torch.empty_like(x), torch.empty_like(v)
        _lib.call("softmax_wsum_backward_kernel_wrapper", x.device, B * C * S, K, x.data_ptr(), v.data_ptr(),
                  dout.data_ptr(), dx.data_ptr(), dv.data_ptr())
        return dx, dv


def softmax_weighted_sum(logits, values):
    """(B,C,S,K), (B,C,S,K) -> (B,C,S) = ``torch.sum(F.softmax(logits, dim=3) * values, dim=3)``."""
    if supported(logits, values):
        return _SoftmaxWeightedSum.apply(logits, values)
    return torch.sum(torch.nn.functional.softmax(logits, dim=3) * values, dim=3)
