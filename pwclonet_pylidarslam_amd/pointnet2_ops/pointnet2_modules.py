"""Set-abstraction and set-upconv modules of PWCLO-Net on the HIP operator stack.

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys as
``P2/pointnet2_modules.py:159-245`` (``PointnetSAModulePWCLONet``) and ``:410-515``
(``PointnetFPModulePWCLONet``).  Centre subtraction uses broadcasting instead of the
reference's materialised ``torch.tile`` copies (same values).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModulePWCLONet(nn.Module):
    """FPS -> gather -> knn -> group -> centre-subtract -> concat -> SharedMLP -> max over K."""

    def __init__(self, mlp: List[int], npoint: int, nsample: int, bn: bool = True):
        super().__init__()
        self.npoint = npoint
        self.nsample = nsample
        mlp_spec = mlp  # the reference mutates the caller's list too (pointnet2_modules.py:170-174)
        if mlp[0] == 0:
            mlp_spec[0] += 3
        mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn, init=torch.nn.init.xavier_uniform_)

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3), (B,mlp[-1],npoint)."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        fps_idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        new_xyz = pointnet2_utils.gather_operation(xyz_flipped, fps_idx).transpose(1, 2).contiguous()
        _, idx_q = pt_utils.knn_point(self.nsample, xyz, new_xyz)
        grouped_xyz = pointnet2_utils.grouping_operation(xyz_flipped, idx_q)
        xyz_diff = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = pointnet2_utils.grouping_operation(features, idx_q)
            new_features = torch.cat((xyz_diff, grouped_features), dim=1)
        else:
            new_features = torch.cat((xyz_diff, grouped_xyz), dim=1)
        new_features = self.mlp_module(new_features)
        new_features = new_features.max(dim=3)[0]  # == max_pool2d(kernel=[1,K]).squeeze(-1)
        return new_xyz, new_features


class PointnetFPModulePWCLONet(nn.Module):
    """Set-upconv: propagate features1 (B,C1,N1) at xyz1 onto the finer xyz2 (B,N2,3)."""

    def __init__(self, *, mlp: List[int], radius: float, nsample: int, post_mlp: List[int],
                 bn: bool = True, use_xyz: bool = True, knn: bool = False,
                 sample_uniformly: bool = False):
        super().__init__()
        self.nsample = nsample
        self.knn = knn
        self.use_xyz = use_xyz
        mlp_spec = mlp
        if use_xyz:
            mlp_spec[0] += 3
        self.mlp = pt_utils.SharedMLP(mlp_spec, bn=bn, init=torch.nn.init.xavier_uniform_)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn, init=torch.nn.init.xavier_uniform_)
        self.grouper = pointnet2_utils.QueryAndGroup(radius, self.nsample, use_xyz=use_xyz)

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        if self.knn:
            _, idx_q = pt_utils.knn_point(self.nsample, xyz1, xyz2)
            new_features = pointnet2_utils.grouping_operation(features1, idx_q)
            grouped_xyz = pointnet2_utils.grouping_operation(xyz1.transpose(1, 2).contiguous(), idx_q)
            xyz_diff = grouped_xyz - xyz2.transpose(1, 2).unsqueeze(-1)
            if self.use_xyz:
                new_features = torch.cat((new_features, xyz_diff), dim=1)
        else:
            new_features = self.grouper(xyz1, xyz2, features1)
        new_features = self.mlp(new_features)
        new_features = new_features.max(dim=3)[0]
        if features2 is not None:
            new_features = torch.cat([new_features, features2], dim=1)
        new_features = self.post_mlp(new_features.unsqueeze(-1))
        return new_features.squeeze(-1)
