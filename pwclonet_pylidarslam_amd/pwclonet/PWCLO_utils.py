"""Quaternion helpers of PWCLO-Net (``PW/PWCLO_utils.py:31-132``), scalar-first.

``warp`` on a point cloud runs the native kernel; the 4-vector products used for pose
composition stay tiny torch expressions.  As in the reference, ``scalar_last`` and ``device``
arguments are accepted and ignored.
"""
import torch

from ..pointnet2_ops import _ext


def _hamilton(a, b):
    """(B,4,N) (x) (B,4,1|N): same component expressions for mul_q_point and mul_point_q
    (PWCLO_utils.py:83-95, 117-129 -- both list the left operand's factors first)."""
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
