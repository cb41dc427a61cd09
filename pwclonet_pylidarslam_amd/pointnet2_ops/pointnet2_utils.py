"""``torch.autograd.Function`` wrappers with the reference's names and signatures
(``P2/pointnet2_utils.py:34-276``) over the HIP extension surface in ``._ext``.

Differentiability matches the reference: FPS / three_nn / ball_query outputs are marked
non-differentiable and their backward returns ``()``; gather / group / three_interpolate send
the gradient to ``features`` only (``grad_out.contiguous()`` first, :97,185,235).
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32 indices (pointnet2_utils.py:36-59)."""
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply

_USE_FPS_CHAIN = os.environ.get("PWCLO_FPS_CHAIN", "1") != "0"
_CHAIN_ATTR = "_pwclo_fps_chain"


def sample_and_gather(xyz, npoint):
    """``furthest_point_sample`` + ``gather_operation`` of the coordinates in one sampler launch:
    xyz (B,N,3) -> new_xyz (B,npoint,3), identical to the two separate calls.

    Sampling chains (csrc/sampling.hip): a pyramid samples each level from the previous level's samples, in
    sampling order, and such a call returns the prefix of its input unless the producing call met an exact
    distance tie.  The first call of a chain records its tie events; the record travels as an attribute of the
    returned tensor, and a later call on THAT tensor (same object, not modified in place since) writes its result
    from the record instead of running the full sampler.  Falls back to the two separate calls whenever that does
    not apply (coordinates that require grad, CPU tensors -> the ext raises as before, ``PWCLO_FPS_CHAIN=0``)."""
    if not (_USE_FPS_CHAIN and xyz.is_cuda and xyz.dtype == torch.float32 and not xyz.requires_grad
            and xyz.dim() == 3 and xyz.is_contiguous() and npoint >= 4):
        idx = furthest_point_sample(xyz, npoint)
        return gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    from .. import fused
    rec = getattr(xyz, _CHAIN_ATTR, None)
    if rec is not None and rec[2] == xyz._version and npoint <= rec[1] and npoint <= xyz.shape[1]:
        _, new_xyz = fused.fps_with_xyz(xyz, npoint, prefix_in=rec[0])
        root = rec
    else:
        tie_iters = npoint // 2
        record = torch.empty((xyz.shape[0], fused.FPS_CHAIN_INTS), dtype=torch.int32, device=xyz.device)
        _, new_xyz = fused.fps_with_xyz(xyz, npoint, tie_out=record, tie_iters=tie_iters)
        root = (record, tie_iters, 0)
    setattr(new_xyz, _CHAIN_ATTR, (root[0], root[1], new_xyz._version))
    return new_xyz


def sample_and_gather_pair(xyz_a, xyz_b, npoint):
    """``sample_and_gather`` of two batches of clouds of the same shape in ONE sampler launch (the sampler runs one
    workgroup per cloud: B clouds use B of the 256 compute units, so the two frames of a pair sampled together take
    the time of one).  Returns ``(new_a, new_b)``, each identical to its own ``sample_and_gather`` call and carrying
    its own rows of the sampling-chain record.  Falls back to two calls when the shapes differ."""
    if xyz_a.shape != xyz_b.shape or not xyz_a.is_cuda or xyz_a.requires_grad or xyz_b.requires_grad:
        return sample_and_gather(xyz_a, npoint), sample_and_gather(xyz_b, npoint)
    B = xyz_a.shape[0]
    both = sample_and_gather(torch.cat((xyz_a, xyz_b), dim=0).contiguous(), npoint)
    rec = getattr(both, _CHAIN_ATTR, None)
    new_a, new_b = both[:B], both[B:]
    if rec is not None:
        setattr(new_a, _CHAIN_ATTR, (rec[0][:B], rec[1], new_a._version))
        setattr(new_b, _CHAIN_ATTR, (rec[0][B:], rec[1], new_b._version))
    return new_a, new_b


_DETERMINISTIC = None


def deterministic_grads(enable=None):
    """Backward of gather / grouping through the atomics-free sorted scatter (``_ext.scatter_grad_
    deterministic``) instead of the reference's fp32 atomics.  Default: env PWCLO_DETERMINISTIC_GRADS
    (0), or whatever ``torch.are_deterministic_algorithms_enabled()`` says."""
    global _DETERMINISTIC
    if enable is not None:
        _DETERMINISTIC = bool(enable)
    if _DETERMINISTIC is not None:
        return _DETERMINISTIC
    import os
    return os.environ.get("PWCLO_DETERMINISTIC_GRADS", "0") != "0" or torch.are_deterministic_algorithms_enabled()


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (pointnet2_utils.py:70-91)."""
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        if deterministic_grads():
            return _ext.scatter_grad_deterministic(grad_out.contiguous(), idx, ctx.n), None
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """-> (dist (B,n,3) = sqrt(dist2), idx (B,n,3)) (pointnet2_utils.py:106-129)."""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx/weight (B,n,3) -> (B,c,n) (pointnet2_utils.py:141-161)."""
        ctx.save_for_backward(idx, weight)
        ctx.m = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m)
        return grad_features, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)
        (pointnet2_utils.py:196-214)."""
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        # the reference returns zeros_like(idx) for the index list (P2/pointnet2_utils.py:237); autograd discards a gradient
        # for an integer tensor either way, so None saves a fill of up to 8 MB per call (44 calls per training step)
        if deterministic_grads():
            return _ext.scatter_grad_deterministic(grad_out.contiguous(), idx, ctx.n), None
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class GroupConcat(Function):
    """``torch.cat([part_0, part_1, ...], dim=1)`` of a shared MLP's input, every part written by its own kernel straight
    into its channel slice of the result, and differentiated straight out of the slice of the incoming gradient:
      "g"   ``grouping_operation(features (B,C,N), idx)``   (P2/pointnet2_modules.py:222-230, 490-500, PW/costvolume.py:134, 172)
      "c"   centre features (B,C,S) tiled over the K neighbours   (PW/costvolume.py:95, 158: torch.tile)
      "geo" the 10-channel geometry encoding of (centre_xyz (B,3,S), src_xyz (B,3,N)) pairs   (PW/costvolume.py:92-105)
      "diff" grouped coordinates relative to their centres, (centre_xyz (B,3,S), src_xyz (B,3,N)) -> 3 channels
            (P2/pointnet2_modules.py:215-218, 485-488: grouping_operation minus the tiled centres)
      "t"   any dense (B,C,S,K) tensor (may be an expanded view): one strided copy.
    The reference writes every part as its own tensor and copies it again in ``torch.cat``.  Same values."""

    ARITY = {"g": 1, "c": 1, "t": 1, "geo": 2, "diff": 2}
    CHANNELS = {"geo": 10, "diff": 3}

    @staticmethod
    def forward(ctx, idx, kinds, *tensors):
        B, S, K = idx.shape
        parts, pos = [], 0
        for kind in kinds:
            ts = tensors[pos:pos + GroupConcat.ARITY[kind]]
            parts.append((kind, pos, ts, GroupConcat.CHANNELS.get(kind) or ts[0].shape[1]))
            pos += len(ts)
        ref = tensors[0]
        out = torch.empty((B, sum(p[3] for p in parts), S, K), dtype=ref.dtype, device=ref.device)
        off = 0
        saved = [idx]
        for kind, _, ts, c in parts:
            if kind == "g":
                _ext.group_points_into(ts[0].contiguous(), idx, out, off)
            elif kind == "c":
                _ext.broadcast_centre_into(ts[0].contiguous(), K, out, off)
            elif kind == "geo":
                cx, sx = ts[0].contiguous(), ts[1].contiguous()
                _ext.geometry_encode_into(cx, sx, idx, out, off)
                saved += [cx, sx]
            elif kind == "diff":
                _ext.xyz_diff_into(ts[0].contiguous(), ts[1].contiguous(), idx, out, off)
            else:
                out[:, off:off + c].copy_(ts[0])          # dense part (may be an expanded view: one strided copy)
            off += c
        ctx.save_for_backward(*saved)
        ctx.parts = [(kind, pos, c, [t.shape[2] for t in ts]) for kind, pos, ts, c in parts]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx = ctx.saved_tensors[0]
        geo_saved = list(ctx.saved_tensors[1:])
        g = grad_out.contiguous()
        det = deterministic_grads()
        inverse = {}
        grads = [None] * sum(GroupConcat.ARITY[p[0]] for p in ctx.parts)
        off = 0
        for kind, pos, c, ns in ctx.parts:
            need = [ctx.needs_input_grad[2 + pos + a] for a in range(GroupConcat.ARITY[kind])]
            if kind == "geo":
                cx, sx = geo_saved.pop(0), geo_saved.pop(0)
                if any(need):
                    n = ns[1]
                    if det and need[1] and n not in inverse:
                        inverse[n] = _ext._inverse_index(idx, n)
                    grads[pos], grads[pos + 1] = _ext.geometry_encode_grad_from(g, off, cx, sx, idx, need[0], need[1],
                                                                                 deterministic=det, inverse=inverse.get(n))
            elif kind == "diff":
                if need[0]:
                    grads[pos] = _ext.broadcast_centre_grad_from(g, off, 3).neg_()
                if need[1]:
                    n = ns[1]
                    if det and n not in inverse:
                        inverse[n] = _ext._inverse_index(idx, n)
                    grads[pos + 1] = _ext.group_points_grad_from(g, off, 3, idx, n, deterministic=det, inverse=inverse.get(n))
            elif not need[0]:
                pass
            elif kind == "g":
                n = ns[0]
                if det and n not in inverse:
                    inverse[n] = _ext._inverse_index(idx, n)
                grads[pos] = _ext.group_points_grad_from(g, off, c, idx, n, deterministic=det, inverse=inverse.get(n))
            elif kind == "c":
                grads[pos] = _ext.broadcast_centre_grad_from(g, off, c)
            else:
                grads[pos] = g[:, off:off + c]
            off += c
        return (None, None) + tuple(grads)


def group_concat(idx, *parts):
    """parts: ``("g", features (B,C,N))``, ``("c", centre features (B,C,S))``, ``("geo" | "diff", centre_xyz (B,3,S),
    src_xyz (B,3,N))`` or ``("t", tensor (B,C,S,K))`` (see ``GroupConcat``); -> (B, sum C, S, K)."""
    return GroupConcat.apply(idx, tuple(p[0] for p in parts), *[t for p in parts for t in p[1:]])


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> (B,npoint,nsample) int32.
        The ext entry point takes (new_xyz, xyz, ...) -- pointnet2_utils.py:265."""
        output = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball-query grouper (pointnet2_utils.py:279-335): (xyz (B,N,3), new_xyz (B,npoint,3),
    features (B,C,N) or None) -> (B, 3+C, npoint, nsample)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features


class QueryAndGroupVoteNet(nn.Module):
    """VoteNet grouper (pointnet2_utils.py:336-418): ``QueryAndGroup`` plus optional normalisation of the offsets by
    the radius, optional return of the grouped offsets, and ``sample_uniformly`` (every ball's distinct members,
    then a random draw among them to fill ``nsample``; also reported as ``unique_cnt``)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if self.sample_uniformly:
            # host loop over balls like the reference (a VoteNet training option, not on PWCLO-Net's path)
            B, M = idx.shape[0], idx.shape[1]
            unique_cnt = torch.zeros((B, M))
            for ball in range(B * M):
                b, r = divmod(ball, M)
                members = torch.unique(idx[b, r])
                unique_cnt[b, r] = members.numel()
                refill = torch.randint(0, members.numel(), (self.nsample - members.numel(),), device=idx.device)
                idx[b, r] = torch.cat((members, members[refill]))
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """pointnet2_utils.py:419-464: one group holding every point -> (B, 3+C, 1, N)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1)
        return grouped_features
