"""Golden values of ONE TRAINING-MODE step of PWCLO-Net -- TEST INFRASTRUCTURE, build container only.

    python -m oracle.gen_train_golden        (writes tests/golden/train_n1024_b2.npz)

The reference trains with batch-statistic BatchNorm (P2/pytorch_utils.py:52-83 SharedMLP -> _BNBase) and
dropout in the pose heads (PW/pose_calculator.py:63-65).  Every other whole-network fixture of this repo is
eval-mode; this one pins the training forward + backward:

  * the imported reference model in ``train()``, with the four ``PoseCalculator`` modules switched to
    ``eval()`` (they hold no BatchNorm; the switch only turns the two ``F.dropout`` calls into the identity,
    whose random stream cannot be reproduced on another device);
  * case n1024_b2 (the inputs of tests/golden/pwclonet_n1024_b2.npz), the oracle's knn (IEEE key, ties ->
    lower index) in place of ``torch.topk``'s unspecified tie order, the C oracle as the CUDA-only ``_ext``;
  * ``_PWCLONetLossModule`` against the seeded ground truth of oracle.gen_grad_golden, ``loss.backward()``
    (slam/training/trainer.py:624-628 without the optimizer step).

The same step is ALSO recorded from the reference model in FLOAT64 (``model.double()``; the index-only ext ops served
by torch gather / scatter_add stand-ins, neighbour lists and FPS from the float32 coordinates): ``grad64.*``,
``loss64``, ``pose64``.  Why: backward through batch-statistic BatchNorm is ill-conditioned in fp32 -- the reference's
own fp32 CPU gradients differ from this float64 evaluation by 3e-4 ... 1.4e-3 of max|g| (printed below; torch's GPU
ops differ from the fp32 CPU values by 6e-4 ... 6e-3), so "equal to the fp32 CPU values to 1e-4" is not a property
any fp32 implementation has.  The GPU test bounds the kernels' error against the float64 values by the reference's own
fp32 error against them.

Recorded: pose_params, loss, the gradients of TRAIN_PARAMS (first to last layer, conv weights AND BatchNorm
affine parameters), the gradient of the loss weights ``s_param``, and the ``running_mean`` / ``running_var``
/ ``num_batches_tracked`` buffers of three BatchNorm layers AFTER the step (momentum update from the batch
statistics, unbiased variance).
"""
import json
import os

import numpy as np
import torch

from oracle import gen_golden, ops, params, ref_import
from oracle.gen_grad_golden import ground_truth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_n1024_b2.npz")
TRAIN_PARAMS = [
    "psa_1.mlp_module.layer0.conv.weight",
    "psa_1.mlp_module.layer0.bn.bn.weight",
    "psa_2.mlp_module.layer2.conv.weight",
    "psa_3.mlp_module.layer1.conv.weight",
    "psa_4.mlp_module.layer2.bn.bn.bias",
    "cost_volume.mlp_convs.layer0.conv.weight",
    "cost_volume.mlp_conv_xyz_1.layer0.conv.weight",
    "cost_volume.mlp3_convs.layer1.conv.weight",
    "flow_feature_encoding.mlp_module.layer1.conv.weight",
    "l4_flow_predictor.mlp_convs.layer0.conv.weight",
    "pose_warp_refinement_3.flow_predictor_mask.mlp_convs.layer1.bn.bn.weight",
    "pose_warp_refinement_2.setupconv_features.mlp.layer0.conv.weight",
    "pose_warp_refinement_2.setupconv_mask.post_mlp.layer0.conv.weight",
    "pose_warp_refinement_1.cost_volume.mlp2_convs.layer0.conv.weight",
    "pose_warp_refinement_1.cost_volume.mlp_conv_xyz_2.layer0.bn.bn.bias",
    "pose_warp_refinement_1.pose_calculator.conv1d_t.conv.weight",
    "pose_calculator_4.conv1d_q.conv.weight",
]
BN_LAYERS = ["psa_1.mlp_module.layer0.bn.bn", "cost_volume.mlp_convs.layer2.bn.bn",
             "pose_warp_refinement_1.flow_predictor_features.mlp_convs.layer1.bn.bn"]


def set_reference_mode(model):
    """train() everywhere, eval() on the dropout-carrying pose heads (they have no BatchNorm)."""
    model.train()
    n = 0
    for name, m in model.named_modules():
        if type(m).__name__ == "PoseCalculator":
            m.eval()
            n += 1
    assert n == 4, n
    return model


def _install_float64_ext(ns):
    """The nine ext ops touch features only through index operations; for float64 tensors serve them with torch
    gather / scatter_add (the float32 calls keep going to the C oracle).  FPS / knn always see float32 coordinates."""
    import sys

    def group_points(points, idx):
        if points.dtype == torch.float32:
            return ops.group_points(points, idx)
        B, C, N = points.shape
        _, S, K = idx.shape
        return torch.gather(points.unsqueeze(2).expand(B, C, S, N), 3, idx.long().unsqueeze(1).expand(B, C, S, K))

    def group_points_grad(g, idx, n):
        if g.dtype == torch.float32:
            return ops.group_points_grad(g, idx, n)
        B, C, S, K = g.shape
        out = torch.zeros(B, C, n, dtype=g.dtype)
        return out.scatter_add_(2, idx.long().reshape(B, 1, S * K).expand(B, C, S * K), g.reshape(B, C, S * K))

    def gather_points(points, idx):
        if points.dtype == torch.float32:
            return ops.gather_points(points, idx)
        B, C, N = points.shape
        return torch.gather(points, 2, idx.long().unsqueeze(1).expand(B, C, idx.shape[1]))

    def gather_points_grad(g, idx, n):
        if g.dtype == torch.float32:
            return ops.gather_points_grad(g, idx, n)
        B, C, M = g.shape
        return torch.zeros(B, C, n, dtype=g.dtype).scatter_add_(2, idx.long().unsqueeze(1).expand(B, C, M), g)

    def fps(points, m):
        return ops.furthest_point_sampling(points.float().contiguous(), m)

    for mod in (sys.modules["pointnet2_ops._ext"], ns.pointnet2_utils._ext):
        mod.group_points, mod.group_points_grad = group_points, group_points_grad
        mod.gather_points, mod.gather_points_grad = gather_points, gather_points_grad
        mod.furthest_point_sampling = fps


def _float64_step(ns, lm, cfg, x1, x2):
    _install_float64_ext(ns)
    model = ref_import.make_reference_model()
    params.fill_state_dict(model.state_dict())
    model = set_reference_mode(model.double())
    loss_mod = lm._PWCLONetLossModule(cfg, lm.Pose("quaternions")).double()
    pose, _ = model(x1.double(), None, x2.double(), None)
    loss, _ = loss_mod(pose, ground_truth(x1.shape[0]).double())
    loss.backward()
    return pose.detach(), loss.detach(), dict(model.named_parameters()), loss_mod.exp_weighting.s_param.grad


def main():
    ns = ref_import.load()
    lm = ref_import.load_loss()
    model = ref_import.make_reference_model()
    params.fill_state_dict(model.state_dict())
    set_reference_mode(model)
    x1, x2 = gen_golden.case_inputs("n1024_b2")
    cfg = ns.DictConfig(mode="supervised", loss_degrees=False, loss_weights=[1.0, 1.0], with_exp_weights=True,
                        init_weights=[0.0, -2.5], loss_option="l2_norm", nb_levels=4, device="cpu", scalar_last=False)
    loss_mod = lm._PWCLONetLossModule(cfg, lm.Pose("quaternions"))
    ref_knn = ns.pytorch_utils.knn_point
    ns.pytorch_utils.knn_point = lambda k, xyz, new_xyz: ops.knn_point(k, xyz.float().contiguous(),
                                                                        new_xyz.float().contiguous())
    try:
        pose, _ = model(x1, None, x2, None)
        loss, _ = loss_mod(pose, ground_truth(x1.shape[0]))
        loss.backward()
        pose64, loss64, named64, gs64 = _float64_step(ns, lm, cfg, x1, x2)
    finally:
        ns.pytorch_utils.knn_point = ref_knn
    named = dict(model.named_parameters())
    missing = [k for k, p in named.items() if p.grad is None]
    assert not missing, missing
    sd = model.state_dict()
    out = {"loss": loss.detach().numpy(), "pose_params": pose.detach().numpy(),
           "grad_s": loss_mod.exp_weighting.s_param.grad.numpy()}
    out.update({"loss64": loss64.numpy(), "pose64": pose64.numpy(), "grad64_s": gs64.numpy()})
    for k in TRAIN_PARAMS:
        out["grad." + k] = named[k].grad.numpy()
        out["grad64." + k] = named64[k].grad.numpy()
        e = (named[k].grad.double() - named64[k].grad).abs().max().item() / named64[k].grad.abs().max().item()
        print("%-78s |g|max %.3e   fp32 reference vs float64 reference: %.2e of max|g|"
              % (k, named[k].grad.abs().max().item(), e))
    for k in BN_LAYERS:
        for s in ("running_mean", "running_var", "num_batches_tracked"):
            out["buf.%s.%s" % (k, s)] = sd["%s.%s" % (k, s)].numpy()
    # scale of every parameter gradient, so that a test can bound ALL of them loosely without the values
    out["all_grad_absmax"] = np.array([named[k].grad.abs().max().item() for k in sorted(named)], dtype=np.float64)
    out["all_grad_l2"] = np.array([named[k].grad.double().norm().item() for k in sorted(named)], dtype=np.float64)
    out["all_grad64_absmax"] = np.array([named64[k].grad.abs().max().item() for k in sorted(named)], dtype=np.float64)
    out["all_grad64_l2"] = np.array([named64[k].grad.norm().item() for k in sorted(named)], dtype=np.float64)
    # the reference's own fp32 error per tensor, in units of max|g64|: the yardstick of the GPU test
    out["all_ref32_err"] = np.array([(named[k].grad.double() - named64[k].grad).abs().max().item()
                                     / max(named64[k].grad.abs().max().item(), 1e-300) for k in sorted(named)])
    out["meta"] = np.array(json.dumps(dict(
        case="n1024_b2", params=TRAIN_PARAMS, bn_layers=BN_LAYERS, gt_seed=515, all_names=sorted(named),
        mode="train (batch-statistic BN), PoseCalculator modules in eval (dropout off)",
        knn="oracle (IEEE key, ties -> lower index)")))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "loss", float(loss))


if __name__ == "__main__":
    main()
