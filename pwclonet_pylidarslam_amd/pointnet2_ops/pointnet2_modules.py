"""Set-abstraction / feature-propagation modules of the ``pointnet2_ops`` package on the HIP operator stack.

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys as
``P2/pointnet2_modules.py``: the PWCLO-Net pair (``PointnetSAModulePWCLONet`` :159-245,
``PointnetFPModulePWCLONet`` :410-515) and the stock PointNet++ modules that are the package's callers of
``ball_query`` and ``three_nn`` / ``three_interpolate`` (``PointnetSAModuleMSG`` / ``PointnetSAModule`` :32-156,
``PointnetFPModule`` :249-326, ``PointnetLFPModuleMSG`` :329-407).  Centre subtraction uses broadcasting instead
of the reference's materialised ``torch.tile`` copies (same values).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


def build_shared_mlp(mlp_spec: List[int], bn: bool = True):
    """pointnet2_modules.py:19-29: plain ``nn.Sequential`` of Conv2d(1x1, bias only without BN) [+ BN] + ReLU."""
    layers = []
    for i in range(1, len(mlp_spec)):
        layers.append(nn.Conv2d(mlp_spec[i - 1], mlp_spec[i], kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(mlp_spec[i]))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class _PointnetSAModuleBase(nn.Module):
    """FPS -> gather -> per scale: grouper -> MLP -> max over the group; scales concatenated."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]
                ) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3) (None for GroupAll),
        (B, sum_k mlps[k][-1], npoint)."""
        new_xyz = None
        if self.npoint is not None:
            xyz_flipped = xyz.transpose(1, 2).contiguous()
            fps_idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz_flipped, fps_idx).transpose(1, 2).contiguous()
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            # (B, mlp[-1], npoint, nsample) -> max over nsample == max_pool2d(kernel=[1,nsample]).squeeze(-1)
            pooled.append(pt_utils.shared_mlp_max(mlp, grouper(xyz, new_xyz, features)))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping: one ball query (radius, nsample) and one MLP per scale (``GroupAll`` when
    ``npoint`` is None).  Like the reference, ``mlps[i][0] += 3`` mutates the caller's lists when ``use_xyz``."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(build_shared_mlp(mlp_spec, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation of ``known_feats`` (B,C2,m) from the three nearest
    known points onto ``unknown`` (B,n,3), concatenated with ``unknow_feats`` (B,C1,n), then a SharedMLP."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:   # global feature: the reference adds a torch.Size to a list here (:314-316), which raises on
            #     current PyTorch; the evident intent is the broadcast below
            interpolated_feats = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)
        else:
            new_features = interpolated_feats
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation (pointnet2_modules.py:329-407): per scale, ball-query groups of
    (xyz1, features1) around xyz2 -> MLP -> max -> [concat features2] -> the SHARED post_mlp; scales concatenated."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int], post_mlp: List[int],
                 bn: bool = True, use_xyz: bool = True, sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(mlp_spec, bn=bn))

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            new_features = pt_utils.shared_mlp_max(mlp, grouper(xyz1, xyz2, features1))     # (B, mlp[-1], N2)
            if features2 is not None:
                new_features = torch.cat([new_features, features2], dim=1)
            outs.append(self.post_mlp(new_features.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)


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

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor], new_xyz: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3), (B,mlp[-1],npoint).
        ``new_xyz`` (not in the reference's signature): the layer's own samples when the caller has already drawn
        them (pointnet2_utils.sample_and_gather_pair samples both frames of a pair in one launch)."""
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        if new_xyz is None:
            new_xyz = pointnet2_utils.sample_and_gather(xyz, self.npoint)   # == furthest_point_sample + gather_operation
        _, idx_q = pt_utils.knn_point(self.nsample, xyz, new_xyz)
        # cat((grouped xyz - centre, grouped features)), every part written by its own kernel into its channel slice
        new_features = pointnet2_utils.group_concat(
            idx_q, ("diff", new_xyz.transpose(1, 2), xyz_flipped), ("g", features if features is not None else xyz_flipped))
        new_features = pt_utils.shared_mlp_max(self.mlp_module, new_features)  # mlp, then max_pool2d(kernel=[1,K])
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
            if self.use_xyz:
                new_features = pointnet2_utils.group_concat(idx_q, ("g", features1),
                                                            ("diff", xyz2.transpose(1, 2), xyz1.transpose(1, 2)))
            else:
                new_features = pointnet2_utils.grouping_operation(features1, idx_q)
        else:
            new_features = self.grouper(xyz1, xyz2, features1)
        new_features = pt_utils.shared_mlp_max(self.mlp, new_features)
        if features2 is not None:
            new_features = torch.cat([new_features, features2], dim=1)
        new_features = self.post_mlp(new_features.unsqueeze(-1))
        return new_features.squeeze(-1)
