"""Pose head, ``PW/pose_calculator.py:20-87``: masked sum over points -> 64->256 -> (q, t)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2_ops.pytorch_utils import Conv1d


class PoseCalculator(nn.Module):
    """(embedding_features (B,C,N), mask (B,C,N)) -> q (B,4[,1]) unit quaternion, t (B,3[,1])."""

    def __init__(self, in_channel: int, out_channel: int, kernel_size: int = 1, padding="valid",
                 activation=None, pose=None, squeeze: bool = True, bn_decay=None):
        super().__init__()
        self.pose = pose
        self.squeeze = squeeze
        xavier = torch.nn.init.xavier_uniform_
        kw = dict(kernel_size=kernel_size, padding=padding, activation=activation, init=xavier)
        self.conv1d_q_t = Conv1d(in_size=in_channel, out_size=out_channel, **kw)
        self.conv1d_q = Conv1d(in_size=out_channel, out_size=4, **kw)
        self.conv1d_t = Conv1d(in_size=out_channel, out_size=3, **kw)

    def forward(self, embedding_features, mask):
        pooled = torch.sum(embedding_features * mask, dim=2, keepdim=True)
        big = self.conv1d_q_t(pooled)
        big_q = F.dropout(big, p=0.5, training=self.training)
        big_t = F.dropout(big, p=0.5, training=self.training)
        q = self.conv1d_q(big_q)
        q = q / (torch.sqrt(torch.sum(q * q, dim=1, keepdim=True) + 1e-10) + 1e-10)
        t = self.conv1d_t(big_t)
        if self.squeeze:
            q, t = q.squeeze(2), t.squeeze(2)
        return q, t
