"""Prediction-module input adapter in front of ``PWCLONet``.

Mirrors ``_PWCLONetPredictionModule`` (``slam/training/prediction_modules.py:103-166``): same
constructor contract (a config with ``device, num_input_channels, sequence_len == 2, num_points,
nb_levels, scalar_last, posenet_config``), same ``forward(data, bn_decay=None) -> (pose_params,
log_dict)`` for ``data`` a dict with keys ``numpy_pc_0`` / ``numpy_pc_1`` or a list of two
``(B, N, C)`` point-major frames, same errors.  The reference slices ``[:, :num_points, :3]``,
permutes to ``(B, 3, N)`` and makes the result contiguous (4 passes over the batch) and the fused
forward would permute back; here an eval-mode net with packed weights (``prepare_fused()``) ingests
the frames in ONE kernel (``ingest_frames_kernel_wrapper``) straight into the point-major batch the
kernels use.  Everything else (training mode, extra feature channels) takes the reference-shaped
route through ``PWCLONet.forward``.
"""
import torch
import torch.nn as nn

from .pwclonet import PWCLONet

NUMPY_PC_KEY = "numpy_pc"        # DatasetLoader.numpy_pc_key(), slam/dataset/configuration.py:67-69


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class PWCLONetPredictionModule(nn.Module):
    def __init__(self, config, pose=None):
        super().__init__()
        self.config = config
        self.pose = pose
        self.device = torch.device(_get(config, "device", "cuda:0"))
        self.num_input_channels = _get(config, "num_input_channels", 3)
        self.sequence_len = _get(config, "sequence_len", 2)
        assert self.sequence_len == 2, "PWCLONet is developed to only accept 2 frames"   # :113
        self.num_points = _get(config, "num_points", 8192)
        self.nb_levels = _get(config, "nb_levels", 4)
        net_cfg = dict(_get(config, "posenet_config", None) or {})
        net_cfg.update(sequence_len=self.sequence_len, num_input_channels=self.num_input_channels,
                       num_points=self.num_points, nb_levels=self.nb_levels, device=str(self.device),
                       scalar_last=_get(config, "scalar_last", False))                      # :117-124
        self.pwclonet = PWCLONet(net_cfg, pose=pose)

    def _frames(self, data):
        if isinstance(data, dict):                                                           # :130-139
            frames = []
            for i in range(self.sequence_len):
                key = "%s_%d" % (NUMPY_PC_KEY, i)
                if key not in data:
                    raise RuntimeError("key `%s` not found in data when running the prediction module" % key)
                frames.append(data[key])
            return frames
        if isinstance(data, list):                                                           # :146-151
            return [data[0], data[1]]
        raise RuntimeError("Input data should be either dict or list")                      # :153-154

    def forward(self, data, bn_decay=None):
        f1, f2 = self._frames(data)
        net = self.pwclonet
        fused = getattr(net, "_fused", None)
        if (fused is not None and not net.training and f1.is_cuda and f1.size(-1) == 3 and f2.size(-1) == 3
                and f1.dtype == torch.float32 and f1.shape == f2.shape):
            pose, inter = fused.forward_frames(f1, f2, min(self.num_points, f1.size(1)), return_intermediates=True)
            return pose, net._fused_log_dict(inter)
        n = self.num_points
        xyz1, xyz2 = f1[:, :n, :3], f2[:, :n, :3]                                           # :141-151
        pts1 = f1[:, :n, 3:] if f1.size(-1) > 3 else None
        pts2 = f2[:, :n, 3:] if f2.size(-1) > 3 else None
        cf = lambda t: t.permute(0, 2, 1).contiguous() if t is not None else None            # :157-160
        return net(cf(xyz1), cf(pts1), cf(xyz2), cf(pts2), bn_decay=bn_decay)
