"""Supervised multi-level pose loss of PWCLO-Net (first slice).

Mirrors ``_PWCLONetLossModule`` and ``ExponentialWeights`` (``slam/training/loss_modules.py:147-197,
325-545``): same constructor contract (``config`` with ``with_exp_weights, init_weights,
loss_weights, loss_option, nb_levels, scalar_last``), ``forward(pred_params (B,4,7), gt_params
(B,7)) -> (loss, log_dict)`` with the same ``log_dict`` keys, the same term order (so that fp32
results match the reference to the last bit on the same device) and the same learnable
``exp_weighting.s_param``.  A few dozen element-wise torch ops on (B,7) tensors: no kernel of its
own; its gradient enters the network through ``pose_params``.
"""
import torch
import torch.nn as nn


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class ExponentialWeights(nn.Module):
    """``loss = sum_i loss_i * exp(-s_i) + s_i`` with learnable ``s`` (loss_modules.py:147-197)."""

    def __init__(self, num_losses, init_weights):
        super().__init__()
        assert len(init_weights) == num_losses
        self.s_param = nn.Parameter(torch.tensor(init_weights), requires_grad=True)
        self.num_losses = num_losses

    def forward(self, list_losses):
        assert len(list_losses) == self.num_losses
        s_params, loss = [], 0.0
        for i in range(self.num_losses):
            s = self.s_param[i]
            loss = loss + (list_losses[i] * torch.exp(-s) + s)
            s_params.append(s.detach())
        return loss, s_params


class PWCLONetLossModule(nn.Module):
    LEVEL_WEIGHTS = (0.2, 0.4, 0.8, 1.6)      # levels 1..4 (loss_modules.py:531)

    def __init__(self, config, pose=None):
        super().__init__()
        self.config = config
        self.pose = pose
        self.exp_weighting = None
        self.weights = None
        self.with_exp_weights = bool(_get(config, "with_exp_weights", True))
        if self.with_exp_weights:
            self.exp_weighting = ExponentialWeights(2, list(_get(config, "init_weights", [0.0, -2.5])))
        else:
            self.weights = list(_get(config, "loss_weights", [1.0, 1.0]))
            assert len(self.weights) == 2
        self.loss_config = _get(config, "loss_option", "l2_norm")
        assert self.loss_config in ("l1", "l2", "l2_norm")
        self.nb_levels = _get(config, "nb_levels", 4)

    # term order of loss_modules.py:355-391
    @staticmethod
    def _l2_norm(x, gt):
        return torch.mean(torch.sqrt(torch.sum((x - gt) * (x - gt), dim=-1, keepdim=True) + 1e-10))

    @staticmethod
    def _trans_loss(x, gt):
        return torch.mean(torch.sqrt((x - gt) * (x - gt) + 1e-10))

    @staticmethod
    def _norm(x):
        return x / (torch.sqrt(torch.sum(x * x, dim=-1, keepdim=True) + 1e-10) + 1e-10)

    def forward(self, pred_params, gt_params):
        levels = [pred_params[:, i, :] for i in range(4)]
        for p in levels:
            assert p.size(1) == 7 and p.size(0) == gt_params.size(0)
        assert gt_params.size(1) == 7
        rot_gt, trans_gt = gt_params[:, 3:], gt_params[:, :3]
        log, level_loss = {}, []
        rot_losses, trans_losses = [], []
        for i, p in enumerate(levels, start=1):
            # the rotation term is the l2 norm whatever loss_option says (loss_modules.py:467-477)
            rot_losses.append(self._l2_norm(self._norm(p[:, 3:]), rot_gt))
            trans_losses.append(self._trans_loss(p[:, :3], trans_gt))
        for i in range(4):
            log["loss_rot_l%d" % (i + 1)] = rot_losses[i]
            log["loss_trans_l%d" % (i + 1)] = trans_losses[i]
        for i in range(4):
            if self.with_exp_weights and self.exp_weighting is not None:
                lvl, s = self.exp_weighting([trans_losses[i], rot_losses[i]])
                log["s_rot_l%d" % (i + 1)] = s[1]
                log["s_trans_l%d" % (i + 1)] = s[0]
            else:
                lvl = trans_losses[i] * self.weights[0] + rot_losses[i] * self.weights[1]
            level_loss.append(lvl)
        for i in range(4):
            log["loss_l%d" % (i + 1)] = level_loss[i]
        loss = 1.6 * level_loss[3] + 0.8 * level_loss[2] + 0.4 * level_loss[1] + 0.2 * level_loss[0]
        log["loss"] = loss
        if self.with_exp_weights and self.exp_weighting is not None:
            sp = self.exp_weighting.s_param.detach()
            # the reference copies these two to the host (loss_modules.py:541-542); a device-to-host copy
            # cannot be captured into a hipGraph, so they stay on the device while a capture is running
            if not (sp.is_cuda and torch.cuda.is_current_stream_capturing()):
                sp = sp.cpu()
            log["s_param_trans"] = sp[0]
            log["s_param_rot"] = sp[1]
        return loss, log
