"""Host-side mirror of ``P2/pytorch_utils.py``: ``knn_point`` on the native kernel, and the
Conv/BN/ReLU building blocks with the reference's ``state_dict`` naming
(``<name>.layer{i}.conv.weight``, ``<name>.layer{i}.bn.bn.{weight,bias,running_mean,...}``).
"""
import os
from typing import List, Tuple

import torch
import torch.nn as nn

from . import _ext
from .. import batchnorm as _hip_bn
from .. import conv1x1 as _hip_conv

_USE_HIP_BN = os.environ.get("PWCLO_HIP_BN", "1") != "0"
# pointwise convolutions on csrc/conv1x1.hip: "all" (default) = every GPU call (no-grad eval blocks run conv + folded
# BatchNorm + ReLU as one kernel), "grad" = only when autograd records the layer (training and gradient checks),
# "0" = torch's convolution everywhere
_USE_HIP_CONV = os.environ.get("PWCLO_HIP_CONV", "all")


def _is_pointwise_on_one_position(conv, x):
    return (isinstance(conv, nn.Conv1d) and x.dim() == 3 and x.shape[2] == 1 and conv.kernel_size == (1,)
            and conv.stride == (1,) and conv.dilation == (1,) and conv.groups == 1
            and (conv.padding == "valid" or conv.padding == (0,)))


def _conv(conv, x):
    if x.is_cuda and _is_pointwise_on_one_position(conv, x):
        # the pose heads (PW/pose_calculator.py:37-39: Conv1d 64->256, 256->4, 256->3 with bias on a (B, C, 1) tensor) are
        # plain matrix products; as ``addmm`` their backward is run-to-run deterministic, which the library's
        # backward-data convolution for this shape is not (tools/train_parity_diag.py: it was the one source of
        # run-to-run differences in the training step's gradients, amplified ~1e4x by the BatchNorm backward chain)
        return torch.nn.functional.linear(x[:, :, 0], conv.weight[:, :, 0], conv.bias).unsqueeze(2)
    if _USE_HIP_CONV != "0" and x.is_cuda and _hip_conv.supported(x, conv) and (
            _USE_HIP_CONV == "all" or (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))):
        return _hip_conv.conv1x1(x, conv.weight)
    return conv(x)


def _nn_distance(pc1, pc2):
    """pytorch_utils.py:12-29: dense (B,N1,N2) matrix of ``sqrt(|p - q|^2 + 1e-8)``.  Kept for callers of the
    package; ``knn_point`` below does not materialise it."""
    diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)
    return torch.sqrt(torch.sum(diff ** 2, dim=-1) + 1e-8)


def knn_point(nsample, xyz, new_xyz):
    """(nsample, xyz (B,N,3), new_xyz (B,S,3)) -> (group_idx, group_idx) with group_idx
    (B,S,nsample) int32, ascending distance.  The reference returns the index tensor in both
    positions (pytorch_utils.py:46-49; every caller discards the first) and so does this.
    Native HIP kernel instead of the dense (B,S,N) distance matrix + torch.topk."""
    idx = _ext.knn_point(nsample, xyz.contiguous(), new_xyz.contiguous())
    return idx, idx


class _BN(nn.Sequential):
    """pytorch_utils.py:86-111: a Sequential holding one BatchNorm under the name ``bn``
    (weight 1, bias 0), so parameters appear as ``...bn.bn.weight``."""

    def __init__(self, in_size, batch_norm, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)

    def train(self, mode=True):
        # the eval path caches the folded (scale, shift) of the running statistics on the BatchNorm module
        # (conv1x1._folded); a mode switch is the moment statistics may have been rewritten behind torch's back
        # (raw-pointer kernels, graph replays): drop it
        self[0].__dict__.pop("_pwclo_folded", None)
        return super().train(mode)

    def forward(self, x):
        bn = self[0]
        # training mode: batch statistics on the HIP kernels (csrc/batchnorm.hip) -- same result as torch's, an
        # order of magnitude faster on the few-channel (B,C,S,K) activations of the grouped MLPs
        if bn.training and _USE_HIP_BN and _hip_bn.supported(x, bn):
            return _hip_bn.batch_norm_train(x, bn)
        return bn(x)


class BatchNorm1d(_BN):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_BN):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class BatchNorm3d(_BN):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, nn.BatchNorm3d, name)


class _ConvBlock(nn.Sequential):
    """pytorch_utils.py:114-167: [bn, act,] conv [, bn, act]; the conv has a bias only when
    there is no batch norm; modules are registered as ``conv`` / ``bn`` / ``activation``."""

    def __init__(self, conv_cls, bn_cls, in_size, out_size, kernel_size, stride, padding, activation,
                 bn, init, bias, preact, name=""):
        super().__init__()
        bias = bias and (not bn)
        conv = conv_cls(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                        bias=bias)
        init(conv.weight)
        if bias:
            nn.init.constant_(conv.bias, 0)
        if preact:
            if bn:
                self.add_module(name + "bn", bn_cls(in_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "conv", conv)
        if not preact:
            if bn:
                self.add_module(name + "bn", bn_cls(out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)


    def forward(self, x):
        # training mode, post-activation order [conv, bn, ReLU]: BatchNorm and ReLU in one pass over the activations
        mods = list(self)
        if (_USE_HIP_BN and len(mods) == 3 and isinstance(mods[1], _BN) and type(mods[2]) is nn.ReLU
                and isinstance(mods[0], (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and mods[1][0].training):
            y = _conv(mods[0], x)
            if _hip_bn.supported(y, mods[1][0]):
                return _hip_bn.batch_norm_train(y, mods[1][0], relu=True)
            return mods[2](mods[1](y))
        if (_USE_HIP_CONV == "all" and x.is_cuda and len(mods) in (2, 3) and isinstance(mods[1], _BN)
                and isinstance(mods[0], (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and (len(mods) == 2 or type(mods[2]) is nn.ReLU)
                and not mods[1][0].training and mods[1][0].track_running_stats and mods[1][0].running_mean is not None
                and not (torch.is_grad_enabled() and (x.requires_grad or mods[0].weight.requires_grad))
                and _hip_conv.supported(x, mods[0])):
            # eval mode, nothing recorded: convolution, folded BatchNorm and ReLU in one kernel
            return _hip_conv.conv1x1_bn_eval(x, mods[0], mods[1][0], relu=len(mods) == 3)
        for m in mods:
            x = _conv(m, x) if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)) else m(x)
        return x


# training-mode stacks: interior BatchNorm + ReLU folded into the next layer's convolution (conv1x1._BNReluConv), the last
# layer's into the stack tail; "0" keeps one BatchNorm pass per layer
_USE_HIP_STACK = os.environ.get("PWCLO_HIP_STACK", "1") != "0"
_USE_CONV_STATS = os.environ.get("PWCLO_CONV_STATS", "1") != "0"     # BatchNorm statistics from the convolution epilogues


def _train_stack(mlp, x, pooled):
    """The whole SharedMLP in training mode with NO normalised activation written: layer 0's convolution, then for every
    following layer ``conv_l(relu(bn_{l-1}(.)))`` as one node, then the last BatchNorm + ReLU (+ max over K when
    ``pooled``).  None when the stack is not of that shape (the caller then runs it layer by layer)."""
    layers = list(mlp)
    if not (_USE_HIP_STACK and _USE_HIP_BN and _USE_HIP_CONV != "0" and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and len(layers) >= 1):
        return None
    blocks = []
    B, P = x.shape[0], x.shape[2] * x.shape[3]
    for layer in layers:
        mods = list(layer) if isinstance(layer, _ConvBlock) else []
        if not (len(mods) == 3 and isinstance(mods[0], nn.Conv2d) and isinstance(mods[1], _BN) and type(mods[2]) is nn.ReLU
                and mods[1][0].training and mods[1][0].momentum is not None
                and (mods[1][0].weight is None) == (mods[1][0].bias is None)
                and _hip_conv.supported_layer(mods[0], B, P, 4)):
            return None
        blocks.append((mods[0], mods[1][0]))
    if blocks[0][0].in_channels != x.shape[1] or (pooled and x.shape[3] not in (4, 8, 16, 32)):
        return None
    last_bn = blocks[-1][1]
    if _USE_CONV_STATS:
        # every convolution sums the batch statistics of its own output in its epilogue: no statistics pass over any
        # activation of the stack
        y, stats = _hip_conv.conv1x1_stats(x, blocks[0][0], blocks[0][1])
        for (_, bn_prev), (conv, bn) in zip(blocks[:-1], blocks[1:]):
            y, stats = _hip_conv.bn_relu_conv(y, bn_prev, conv, stats=stats, next_bn=bn)
        return (_hip_bn.batch_norm_train_relu_max(y, last_bn, stats=stats) if pooled
                else _hip_bn.batch_norm_train(y, last_bn, relu=True, stats=stats))
    y = _hip_conv.conv1x1(x, blocks[0][0].weight)
    for (_, bn_prev), (conv, _) in zip(blocks[:-1], blocks[1:]):
        y = _hip_conv.bn_relu_conv(y, bn_prev, conv)
    return _hip_bn.batch_norm_train_relu_max(y, last_bn) if pooled else _hip_bn.batch_norm_train(y, last_bn, relu=True)


def shared_mlp_max(mlp, x):
    """``mlp(x).max(dim=3)[0]`` for a SharedMLP ``mlp`` on x (B, C, S, K) -- the tail every grouped stack of the
    reference ends with (P2/pointnet2_modules.py: SA, set-upconv).  In training mode on the GPU the last layer's
    BatchNorm, ReLU and the max run as one op (batchnorm.batch_norm_train_relu_max): same values, the largest
    activation of the stack and its gradient are never written."""
    layers = list(mlp)
    mods = list(layers[-1]) if layers and isinstance(layers[-1], _ConvBlock) else []
    if mods and isinstance(mods[1] if len(mods) > 1 else None, _BN) and mods[1][0].training:
        out = _train_stack(mlp, x, pooled=True)
        if out is not None:
            return out
    if (_USE_HIP_BN and x.is_cuda and len(mods) == 3 and isinstance(mods[0], nn.Conv2d) and isinstance(mods[1], _BN)
            and type(mods[2]) is nn.ReLU and mods[1][0].training):
        for layer in layers[:-1]:
            x = layer(x)
        y = _conv(mods[0], x)
        if _hip_bn.supported_maxk(y, mods[1][0]):
            return _hip_bn.batch_norm_train_relu_max(y, mods[1][0])
        return mods[2](mods[1](y)).max(dim=3)[0]
    if (_USE_HIP_CONV == "all" and x.is_cuda and len(mods) == 3 and isinstance(mods[0], nn.Conv2d)
            and isinstance(mods[1], _BN) and type(mods[2]) is nn.ReLU and not mods[1][0].training
            and mods[1][0].track_running_stats and mods[1][0].running_mean is not None and not torch.is_grad_enabled()):
        # eval mode, nothing recorded: the last layer's convolution, folded BatchNorm, ReLU and the max in one kernel
        for layer in layers[:-1]:
            x = layer(x)
        if x.dim() == 4 and x.shape[3] in (4, 8, 16, 32) and _hip_conv.supported(x, mods[0]):
            return _hip_conv.conv1x1_bn_eval_maxk(x, mods[0], mods[1][0], relu=True)
        return layers[-1](x).max(dim=3)[0]
    return mlp(x).max(dim=3)[0]


class Conv1d(_ConvBlock):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1,
                 padding=0, activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False, name: str = ""):
        super().__init__(nn.Conv1d, BatchNorm1d, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class Conv2d(_ConvBlock):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Tuple[int, int] = (1, 1),
                 stride: Tuple[int, int] = (1, 1), padding=(0, 0), activation=nn.ReLU(inplace=True),
                 bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = ""):
        super().__init__(nn.Conv2d, BatchNorm2d, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class Conv3d(_ConvBlock):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Tuple[int, int, int] = (1, 1, 1),
                 stride: Tuple[int, int, int] = (1, 1, 1), padding=(0, 0, 0), activation=nn.ReLU(inplace=True),
                 bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = ""):
        super().__init__(nn.Conv3d, BatchNorm3d, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class FC(nn.Sequential):
    """pytorch_utils.py:272-307: [bn, act,] Linear [, bn, act]; the Linear has a bias only without batch norm."""

    def __init__(self, in_size: int, out_size: int, *, activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=None, preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(in_size))
            if activation is not None:
                self.add_module(name + "activation", activation)
        self.add_module(name + "fc", fc)
        if not preact:
            if bn:
                self.add_module(name + "bn", BatchNorm1d(out_size))
            if activation is not None:
                self.add_module(name + "activation", activation)


class SharedMLP(nn.Sequential):
    """pytorch_utils.py:52-83: ``layer{i}`` = Conv2d(1x1) [+ BN] + ReLU over (B,C,S,K)."""

    def forward(self, x):
        if self.training and x.is_cuda:
            out = _train_stack(self, x, pooled=False)
            if out is not None:
                return out
        return super().forward(x)

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True),
                 preact: bool = False, first: bool = False, name: str = "",
                 init=nn.init.kaiming_normal_):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=plain and bn,
                                   activation=activation if plain else None, preact=preact, init=init))


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """pytorch_utils.py:319-349."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.last_momentum = self.lmbd(0)
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.last_momentum = self.lmbd(epoch)
        self.model.apply(self.setter(self.lmbd(epoch)))
