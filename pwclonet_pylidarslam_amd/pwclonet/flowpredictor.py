"""Embedding-feature / embedding-mask predictor, ``PW/flowpredictor.py:15-83``."""
import torch
import torch.nn as nn

from ..pointnet2_ops import pytorch_utils as pt_utils


class FlowPredictor(nn.Module):
    """cat(points_f1, cost_volume[, upsampled_feat]) (B,C,N) -> SharedMLP -> (B,mlp[-1],N)."""

    def __init__(self, in_channel, mlp, bn_decay=None):
        super().__init__()
        self.in_channel = [in_channel]
        mlp_spec = [in_channel] + mlp
        self.mlp_convs = pt_utils.SharedMLP(mlp_spec, bn=True, init=torch.nn.init.xavier_uniform_)
        self.out_channel = mlp_spec[-1]

    def forward(self, points_f1, cost_volume, upsampled_feat=None):
        if points_f1 is None:
            x = cost_volume
        elif upsampled_feat is not None:
            x = torch.cat((points_f1, cost_volume, upsampled_feat), dim=1)
        else:
            x = torch.cat((points_f1, cost_volume), dim=1)
        return self.mlp_convs(x.unsqueeze(3)).squeeze(3)
