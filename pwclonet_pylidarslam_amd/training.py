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


def set_reference_train_mode(net, dropout=True):
    """``net.train()`` as the reference's trainer leaves the model (batch-statistic BatchNorm everywhere,
    P2/pytorch_utils.py:52-83).  ``dropout=False`` additionally puts the four ``PoseCalculator`` heads -- which
    hold no BatchNorm, only the two ``F.dropout`` calls of PW/pose_calculator.py:63-65 -- into ``eval()``: the mode
    the training parity fixtures are recorded in (a dropout stream is not reproducible across devices)."""
    net.train()
    if not dropout:
        heads = [m for m in net.modules() if type(m).__name__ == "PoseCalculator"]
        assert len(heads) == 4, len(heads)
        for m in heads:
            m.eval()
    return net


class TrainStep:
    """One training step of the data-parallel unit -- ``zero_grad -> forward -> loss -> backward -> optimizer.step``
    (slam/training/trainer.py:624-628) -- launched eagerly or replayed as ONE hipGraph (``graph=True``; single
    process only: DDP's bucketed all-reduce is not captured here).  ``step()`` returns the loss tensor (static under
    the graph: read it before the next replay)."""

    def __init__(self, model, optimizer, xyz_f1, xyz_f2, gt_params, graph=False, warmup=3):
        self.model, self.opt = model, optimizer
        self.args = (xyz_f1, xyz_f2, gt_params)
        self.graph = None
        if graph:
            dev = xyz_f1.device
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):                 # allocator / autograd warm-up outside the capture
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.static_loss, _pose, _log = self.model(*self.args)
                self.static_loss.backward()
                self.opt.step()

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss, _pose, _log = self.model(*self.args)
        loss.backward()
        self.opt.step()
        return loss

    def step(self):
        if self.graph is None:
            return self._eager()
        self.graph.replay()                             # gradients are overwritten in place by the replay
        return self.static_loss
