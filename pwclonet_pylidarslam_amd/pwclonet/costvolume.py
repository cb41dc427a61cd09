"""Attentive cost volume (double attentive embedding), ``PW/costvolume.py:19-190``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2_ops import pointnet2_utils as pointutils
from ..pointnet2_ops import pytorch_utils as pt_utils
from ..softmax_wsum import softmax_weighted_sum


class CostVolume(nn.Module):
    """(warped_xyz (B,3,S), warped_points (B,C1,S), f2_xyz (B,3,N), f2_points (B,C2,N))
    -> (B, mlp2[-1], S).  Requires mlp1[-1] == mlp2[-1]."""

    def __init__(self, nsample, nsample_q, in_channel1, in_channel2, mlp1, mlp2):
        super().__init__()
        self.nsample = nsample
        self.nsample_q = nsample_q
        self.in_channel = [in_channel1, in_channel2, 10]
        xavier = torch.nn.init.xavier_uniform_
        mlp1_spec = [in_channel1 + in_channel2 + 10] + mlp1
        self.mlp_convs = pt_utils.SharedMLP(mlp1_spec, bn=True, init=xavier)
        self.mlp_conv_xyz_1 = pt_utils.SharedMLP([10, mlp1[-1]], bn=True, init=xavier)
        self.mlp_conv_xyz_2 = pt_utils.SharedMLP([10, mlp1[-1]], bn=True, init=xavier)
        self.mlp2_convs = pt_utils.SharedMLP([mlp1_spec[-1] * 2] + mlp2, bn=True, init=xavier)
        mlp3_spec = [mlp1_spec[-1] * 2 + in_channel1] + mlp2
        self.mlp3_convs = pt_utils.SharedMLP(mlp3_spec, bn=True, init=xavier)
        self.out_channel = mlp3_spec[-1]

    def forward(self, warped_xyz, warped_points, f2_xyz, f2_points):
        warped_xyz_t = warped_xyz.permute(0, 2, 1).contiguous()
        f2_xyz_t = f2_xyz.permute(0, 2, 1).contiguous()
        kq, k = self.nsample_q, self.nsample
        gc = pointutils.group_concat

        # first aggregate: frame-2 neighbours of every (warped) frame-1 point.  cat((geometry, tiled centre features,
        # grouped frame-2 features)) and the geometry alone, every part written by its own kernel (no tile / sub / square
        # / sum / sqrt / cat chain, forward or backward)
        _, idx_q = pt_utils.knn_point(kq, f2_xyz_t, warped_xyz_t)
        feat = self.mlp_convs(gc(idx_q, ("geo", warped_xyz, f2_xyz), ("c", warped_points), ("g", f2_points)))
        enc = self.mlp_conv_xyz_1(gc(idx_q, ("geo", warped_xyz, f2_xyz)))
        # softmax over the neighbours + weighted sum (costvolume.py:139-141) as one kernel each way on the GPU
        first = softmax_weighted_sum(self.mlp2_convs(torch.cat((enc, feat), dim=1)), feat)

        # second aggregate: frame-1 neighbours of every frame-1 point
        _, idx = pt_utils.knn_point(k, warped_xyz_t, warped_xyz_t)
        c_pts = pointutils.grouping_operation(first.contiguous(), idx)
        enc2 = self.mlp_conv_xyz_2(gc(idx, ("geo", warped_xyz, warped_xyz)))
        return softmax_weighted_sum(self.mlp3_convs(gc(idx, ("t", enc2), ("c", warped_points), ("t", c_pts))), c_pts)   # :181-183
