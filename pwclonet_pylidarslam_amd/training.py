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

    def forward(self, xyz_f1, xyz_f2, gt_params, samples=None):
        if samples is None:
            pose, _ = self.pwclonet(xyz_f1, None, xyz_f2, None)
        else:
            pose, _ = self.pwclonet(xyz_f1, None, xyz_f2, None, samples=samples)
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
    the graph: read it before the next replay).

    ``sample_ahead=True``: the furthest-point sampling of the four pyramid levels -- a function of the input
    coordinates only, the longest serial kernel chain of the step (one workgroup per cloud: 1.9 + up to 0.7 ms on a
    quarter of the chip) -- is drawn for the NEXT batch on a second stream WHILE the current batch's step runs, the way
    a data loader prefetches: ``step(next_batch=(xyz_f1, xyz_f2, gt))`` trains on the batch loaded last (the
    constructor's at first), samples ``next_batch`` beside it and makes it current; ``step()`` keeps the same batch.
    Every step still runs one sampling chain and one forward / backward / optimizer step; the values are those of the
    plain step (same samples, bit for bit; tests/test_gpu_train.py).  Under ``graph=True`` the sampler is a second
    hipGraph replayed on the side stream.  OPT-IN, and measured (tools/sample_ahead_probe.py, profiles/r03): the step
    graph without its sampler replays in 25.2 ms instead of 27.7, but on this runtime a graph replayed on the default
    stream does not overlap with work of another stream (25.2 + 2.6 = 27.7 ms again), the same graph replayed on a
    non-default stream takes 56 ms, and a forked branch inside ONE graph 61 ms; eager kernels of two streams do overlap
    (20 matmuls + the sampler: 19.3 ms against 18.3 + 2.6), but the eager step is host-bound.  So today this buys
    nothing on one GPU; it is the hook a loader-side sampler needs."""

    def __init__(self, model, optimizer, xyz_f1, xyz_f2, gt_params, graph=False, warmup=3, sample_ahead=False):
        self.model, self.opt = model, optimizer
        self.args = (xyz_f1, xyz_f2, gt_params)
        self.graph = self.sample_graph = None
        self.samples = self.next_samples = None
        dev = xyz_f1.device
        if sample_ahead:
            unit = model.module if hasattr(model, "module") else model
            self.net = unit.pwclonet
            self.side = torch.cuda.Stream(device=dev)
            self.next_args = tuple(t.clone() for t in self.args)
            self.samples = self.net.sample_pyramid(xyz_f1, xyz_f2)            # prologue: the first batch's own samples
            self.samples = tuple([t.clone() for t in lv] for lv in self.samples)
        if graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):                 # allocator / autograd warm-up outside the capture
                    self._eager()
                    if sample_ahead:
                        self._sample_next()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.static_loss, _pose, _log = self._forward()
                self.static_loss.backward()
                self.opt.step()
            if sample_ahead:
                self.sample_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.sample_graph):
                    self._sample_next()
                torch.cuda.synchronize(dev)

    def _forward(self):
        if self.samples is None:
            return self.model(*self.args)
        return self.model(*self.args, samples=self.samples)

    def _sample_next(self):
        nxt = self.net.sample_pyramid(self.next_args[0], self.next_args[1])
        if self.next_samples is None:
            self.next_samples = tuple([t.clone() for t in lv] for lv in nxt)
        else:
            for dst, src in zip(self.next_samples, nxt):
                torch._foreach_copy_(dst, src)

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss, _pose, _log = self._forward()
        loss.backward()
        self.opt.step()
        return loss

    def step(self, next_batch=None):
        if self.samples is None:
            if next_batch is not None:
                raise ValueError("next_batch needs TrainStep(sample_ahead=True); load a batch by copying into the tensors "
                                 "the step was built on")
            if self.graph is None:
                return self._eager()
            self.graph.replay()                         # gradients are overwritten in place by the replay
            return self.static_loss
        cur = torch.cuda.current_stream(self.args[0].device)
        if next_batch is not None:
            torch._foreach_copy_(list(self.next_args), list(next_batch))
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):              # the next batch's samples, beside this batch's step
            if self.sample_graph is not None:
                self.sample_graph.replay()
            else:
                self._sample_next()
        if self.graph is None:
            loss = self._eager()
        else:
            self.graph.replay()
            loss = self.static_loss
        cur.wait_stream(self.side)
        if next_batch is not None:
            torch._foreach_copy_(list(self.args), list(self.next_args))
        for dst, src in zip(self.samples, self.next_samples):
            torch._foreach_copy_(dst, src)
        return loss
