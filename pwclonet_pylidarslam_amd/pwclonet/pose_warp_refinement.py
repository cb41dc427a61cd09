"""Pose warp-refinement level, ``PW/pose_warp_refinement.py:25-158``."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2_ops.pointnet2_modules import PointnetFPModulePWCLONet
from . import PWCLO_utils as pwclo
from .costvolume import CostVolume
from .flowpredictor import FlowPredictor
from .pose_calculator import PoseCalculator


class PoseWarpRefinement(nn.Module):
    """Set-upconv x2 -> warp by the coarse pose -> cost volume -> embedding / mask refinement ->
    residual pose -> composition.  Tensors: xyz_* (B,3,N*), points_* (B,C*,N*)."""

    def __init__(self, in_channel_f1: int, in_channel_f2: int, in_channel_f1_prev: int,
                 in_channel_mask: int, knn: bool = False, radius: float = 0.0,
                 last_pose_estimation: bool = False, pose=None, device: str = "cpu",
                 scalar_last: bool = True):
        super().__init__()
        if (not knn) and (radius == 0.0):
            raise RuntimeError("PoseWarpRefinement: when `knn` is set to False, `radius` should be "
                               "precised.")
        self.pose = pose
        self.device = device
        self.scalar_last = scalar_last
        self.last_pose_estimation = last_pose_estimation
        self.in_channel = [in_channel_f1, in_channel_f2, in_channel_f1_prev]
        up = dict(nsample=8, post_mlp=[64 + in_channel_f1, 64], radius=radius * 0.2, knn=True,
                  use_xyz=True, bn=True)
        self.setupconv_features = PointnetFPModulePWCLONet(mlp=[in_channel_f1_prev, 128, 64], **up)
        up["post_mlp"] = [64 + in_channel_f1, 64]
        self.setupconv_mask = PointnetFPModulePWCLONet(mlp=[in_channel_mask, 128, 64], **up)
        self.cost_volume = CostVolume(nsample=4, nsample_q=6, in_channel1=in_channel_f1,
                                      in_channel2=in_channel_f2, mlp1=[128, 64, 64], mlp2=[128, 64])
        self.flow_predictor_features = FlowPredictor(in_channel=in_channel_f1 + 64 + 64, mlp=[128, 64])
        if not self.last_pose_estimation:
            self.flow_predictor_mask = FlowPredictor(in_channel=in_channel_f1 + 64 + 64, mlp=[128, 64])
        self.pose_calculator = PoseCalculator(in_channel=64, out_channel=256, kernel_size=1,
                                              padding="valid", activation=None, pose=pose,
                                              squeeze=False)
        self.out_channel = [4, 3, 64]

    def forward(self, xyz_f1, points_f1, xyz_f2, points_f2, xyz_f1_prev, points_f1_prev,
                embedding_mask_prev, q_prev, t_prev):
        B = xyz_f1.size(0)
        q_coarse = q_prev.reshape(B, 4, 1)
        t_coarse = t_prev.reshape(B, 3, 1)
        xyz_f1_t = xyz_f1.permute(0, 2, 1).contiguous()
        xyz_prev_t = xyz_f1_prev.permute(0, 2, 1).contiguous()

        coarse_features = self.setupconv_features(xyz_f1_t, xyz_prev_t, points_f1, points_f1_prev)
        coarse_masks = self.setupconv_mask(xyz_f1_t, xyz_prev_t, points_f1, embedding_mask_prev)
        warped_xyz_f1 = pwclo.warp(xyz_f1, q_coarse, t_coarse)
        residual = self.cost_volume(warped_xyz_f1, points_f1, xyz_f2, points_f2)
        embedding_features = self.flow_predictor_features(points_f1, residual, coarse_features)
        if not self.last_pose_estimation:
            embedding_mask = self.flow_predictor_mask(coarse_masks, embedding_features, points_f1)
        else:
            embedding_mask = coarse_masks
        q_det, t_det = self.pose_calculator(embedding_features, F.softmax(embedding_mask, dim=2))
        q = pwclo.mul_point_q(q_det, q_coarse).squeeze(2)        # pose_warp_refinement.py:139
        t = pwclo.warp(t_coarse, q_det, t_det).squeeze(2)        # :148
        return q, t, embedding_features, embedding_mask
