"""Quaternion helpers of PWCLO-Net (``PW/PWCLO_utils.py:31-132``), scalar-first.

``warp`` on a point cloud runs the native kernel; the 4-vector products used for pose
composition stay tiny torch expressions.  As in the reference, ``scalar_last`` and ``device``
arguments are accepted and ignored.
"""
import torch

from ..pointnet2_ops import _ext


class _Hamilton(torch.autograd.Function):
    """The Hamilton product on one kernel (csrc/warp.hip hamilton_kernel) instead of 16 products, 12 sums and a stack:
    forward values are the torch expression's bit for bit; the gradients are products with conjugates
    (dA = dC (x) conj(B), dB = conj(A) (x) dC, summed over N for a broadcast operand), computed by the same op -- so
    it is differentiable to any order."""

    @staticmethod
    def forward(ctx, a, b, conj_a, conj_b):
        a, b = a.contiguous(), b.contiguous()
        B, n = a.shape[0], max(a.shape[2], b.shape[2])
        out = torch.empty((B, 4, n), dtype=torch.float32, device=a.device)
        from .. import _lib
        _lib.call("hamilton_product_kernel_wrapper", a.device, B, n, a.shape[2], b.shape[2], int(conj_a), int(conj_b),
                  a.data_ptr(), b.data_ptr(), out.data_ptr())
        ctx.save_for_backward(a, b)
        ctx.conj = (bool(conj_a), bool(conj_b))
        return out

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ca, cb = ctx.conj
        da = db = None
        if ctx.needs_input_grad[0]:
            # C = A' (x) B' with A' = A or conj(A): dA' = dC (x) conj(B'); dA = conj(dA') when A' = conj(A)
            da = _Hamilton.apply(dc, b, False, not cb)
            if ca:
                da = torch.cat((da[:, :1], -da[:, 1:]), dim=1)
            if a.shape[2] == 1 and da.shape[2] != 1:
                da = da.sum(dim=2, keepdim=True)
        if ctx.needs_input_grad[1]:
            db = _Hamilton.apply(a, dc, not ca, False)
            if cb:
                db = torch.cat((db[:, :1], -db[:, 1:]), dim=1)
            if b.shape[2] == 1 and db.shape[2] != 1:
                db = db.sum(dim=2, keepdim=True)
        return da, db, None, None


def _hamilton(a, b):
    """(B,4,N) (x) (B,4,1|N): same component expressions for mul_q_point and mul_point_q
    (PWCLO_utils.py:83-95, 117-129 -- both list the left operand's factors first)."""
    if a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 3 and b.dim() == 3 \
            and a.shape[1] == 4 and b.shape[1] == 4 and a.shape[0] == b.shape[0] \
            and (a.shape[2] == b.shape[2] or a.shape[2] == 1 or b.shape[2] == 1):
        return _Hamilton.apply(a, b, False, False)
    a0, a1, a2, a3 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    b0, b1, b2, b3 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack((a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3,
                        a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2,
                        a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
                        a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0), dim=1)


def mul_q_point(q, points, scalar_last: bool = False):
    """q (B,4[,1]) (x) points (B,4,N) -> (B,4,N)."""
    return _hamilton(q.reshape(points.size(0), 4, 1), points)


def mul_point_q(points, q, scalar_last: bool = False):
    """points (B,4,N) (x) q (B,4[,1]) -> (B,4,N)."""
    return _hamilton(points, q.reshape(points.size(0), 4, 1))


def inv_q(q, device=None, scalar_last: bool = False):
    """conj(q) / (|q|^2 + 1e-10); q (B,4)."""
    q2 = torch.sum(q * q, dim=-1, keepdim=True) + 1e-10
    # conj(q): same values as the reference's `q * [1,-1,-1,-1]` (PWCLO_utils.py:36) without building a
    # device tensor from host data (a host->device copy cannot be captured into a hipGraph)
    conj = torch.cat((q[..., :1], -q[..., 1:]), dim=-1)
    return conj / q2


def warp(xyz, q, t, device=None, scalar_last: bool = False):
    """xyz (B,3,N), q (B,4,1), t (B,3,1) -> q (x) (0,xyz) (x) q^-1 + t, (B,3,N).
    Native kernel when no gradient is needed; differentiable torch expression otherwise."""
    if not (torch.is_grad_enabled() and (xyz.requires_grad or q.requires_grad or t.requires_grad)):
        return _ext.quat_warp(xyz.contiguous(), q, t)
    B, _, N = xyz.shape
    qi = inv_q(q.reshape(B, 4))
    p = torch.cat((torch.zeros(B, 1, N, dtype=xyz.dtype, device=xyz.device), xyz), dim=1)
    r = mul_point_q(mul_q_point(q, p), qi)
    return r[:, 1:, :] + t.reshape(B, 3, 1)
