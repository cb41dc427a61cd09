"""The unit that data-parallel training replicates: network + loss in ONE module.

The reference's trainer optimises the prediction module's parameters together with the loss module's
learnable weights (``slam/training/trainer.py:268`` builds both; ``loss_modules.py:147-197``: the
``ExponentialWeights.s_param`` pair).  Under ``DistributedDataParallel`` every parameter the optimiser
steps must take part in the gradient all-reduce, otherwise each rank -- which sees different frame pairs --
drifts to its own loss weights and then to its own network.  ``PWCLONetWithLoss`` owns both, so that
``ddp(PWCLONetWithLoss(net, loss))`` reduces the 775 068 network gradients AND the 2 loss-weight gradients
("775 068 + 2 fp32 values = 3.10 MB per step") in one bucket.
"""
import torch
import torch.nn as nn


class PWCLONetWithLoss(nn.Module):
    """``forward(xyz_f1 (B,3,N), xyz_f2 (B,3,N), gt_params (B,7)) -> (loss, pose_params (B,4,7), log_dict)``."""

    def __init__(self, net, loss_module):
        super().__init__()
        self.pwclonet = net
        self.loss_module = loss_module

    def forward(self, xyz_f1, xyz_f2, gt_params):
        pose, _ = self.pwclonet(xyz_f1, None, xyz_f2, None)
        loss, log = self.loss_module(pose, gt_params)
        return loss, pose, log


def ddp_wrap(model, device=None, process_group=None):
    """``DistributedDataParallel`` over RCCL with the settings SURVEY.md section 8e derives for this model:
    BN buffers are NOT broadcast (the reference has no cross-rank BN), every parameter receives a gradient
    (``find_unused_parameters=False``), and the whole 3.1 MB gradient set travels as one bucket (the step is
    latency-bound over xGMI: a single all-reduce, not DDP's default 1 MB first bucket + remainder)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    kw = dict(broadcast_buffers=False, find_unused_parameters=False, bucket_cap_mb=16, process_group=process_group)
    if device is not None and device.type == "cuda":
        kw["device_ids"] = [device.index if device.index is not None else torch.cuda.current_device()]
    ddp = DDP(model, **kw)
    return ddp


def gradient_bucket_values(model):
    """Number of fp32 values one step's all-reduce carries (parameters that require grad)."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
