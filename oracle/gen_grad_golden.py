"""Golden gradients of the whole PWCLO-Net backward -- TEST INFRASTRUCTURE, build container only.

    python -m oracle.gen_grad_golden        (writes tests/golden/grad_n1024_b2.npz)

Imports the reference model and loss module on CPU (oracle.ref_import; the C oracle stands in for the
CUDA-only ext, so the reference's own autograd.Functions call ``group_points_grad`` /
``gather_points_grad`` of the oracle), runs the n1024_b2 case in eval mode with the oracle's knn
(deterministic ties), takes ``_PWCLONetLossModule`` against a seeded ground truth and records the
loss and the gradients of a few parameters from the first to the last layer of the network.
"""
import json
import os

import numpy as np
import torch

from oracle import gen_golden, ops, params, ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grad_n1024_b2.npz")
PARAMS = ["psa_1.mlp_module.layer0.conv.weight", "psa_3.mlp_module.layer1.conv.weight",
          "cost_volume.mlp_convs.layer0.conv.weight", "cost_volume.mlp3_convs.layer1.conv.weight",
          "pose_warp_refinement_2.setupconv_features.mlp.layer0.conv.weight",
          "pose_warp_refinement_1.cost_volume.mlp2_convs.layer0.conv.weight",
          "pose_warp_refinement_1.pose_calculator.conv1d_t.conv.weight", "pose_calculator_4.conv1d_q.conv.weight"]


def ground_truth(batch):
    g = torch.Generator().manual_seed(515)
    gt = torch.randn(batch, 7, generator=g) * 0.2
    gt[:, 3:] = torch.nn.functional.normalize(gt[:, 3:] + torch.tensor([1.0, 0, 0, 0]), dim=1)
    return gt


def main():
    ns = ref_import.load()
    lm = ref_import.load_loss()
    model = ref_import.make_reference_model().eval()
    params.fill_state_dict(model.state_dict())
    x1, x2 = gen_golden.case_inputs("n1024_b2")
    cfg = ns.DictConfig(mode="supervised", loss_degrees=False, loss_weights=[1.0, 1.0], with_exp_weights=True,
                        init_weights=[0.0, -2.5], loss_option="l2_norm", nb_levels=4, device="cpu", scalar_last=False)
    loss_mod = lm._PWCLONetLossModule(cfg, lm.Pose("quaternions"))
    ref_knn = ns.pytorch_utils.knn_point
    ns.pytorch_utils.knn_point = lambda k, xyz, new_xyz: ops.knn_point(k, xyz.contiguous(), new_xyz.contiguous())
    try:
        pose, _ = model(x1, None, x2, None)
        loss, _ = loss_mod(pose, ground_truth(x1.shape[0]))
        loss.backward()
    finally:
        ns.pytorch_utils.knn_point = ref_knn
    named = dict(model.named_parameters())
    out = {"loss": loss.detach().numpy(), "pose_params": pose.detach().numpy(),
           "grad_s": loss_mod.exp_weighting.s_param.grad.numpy()}
    for k in PARAMS:
        out["grad." + k] = named[k].grad.numpy()
        print("%-72s |g|max %.3e" % (k, named[k].grad.abs().max().item()))
    out["meta"] = np.array(json.dumps(dict(case="n1024_b2", params=PARAMS, gt_seed=515, mode="eval (BN running stats)",
                                           knn="oracle (IEEE key, ties -> lower index)")))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "loss", float(loss))


if __name__ == "__main__":
    main()
