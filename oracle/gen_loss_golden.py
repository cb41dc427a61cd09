"""Golden vectors for the PWCLO-Net loss module -- TEST INFRASTRUCTURE, build container only.

    python -m oracle.gen_loss_golden        (writes tests/golden/loss_cases.npz)

Imports the reference's ``_PWCLONetLossModule`` (slam/training/loss_modules.py) on CPU with the stubs
of ``oracle.ref_import`` and records, for seeded (pred, gt) pairs and both weighting modes, the
loss, every scalar of its ``log_dict`` and the gradients w.r.t. ``pred_params`` and ``s_param``.
"""
import json
import os

import numpy as np
import torch

from oracle import ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loss_cases.npz")


def inputs(seed, batch):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(batch, 4, 7, generator=g) * 0.5
    pred[:, :, 3] += 1.0                                   # quaternions near identity, un-normalised
    gt = torch.randn(batch, 7, generator=g) * 0.3
    gt[:, 3:] = torch.nn.functional.normalize(gt[:, 3:] + torch.tensor([1.0, 0, 0, 0]), dim=1)
    return pred, gt


def main():
    lm = ref_import.load_loss()
    ns = ref_import.load()
    out, meta = {}, {"cases": []}
    for seed, batch, exp in ((1, 5, True), (2, 32, True), (3, 7, False)):
        cfg = ns.DictConfig(mode="supervised", loss_degrees=False, loss_weights=[1.0, 0.5], with_exp_weights=exp,
                            init_weights=[0.0, -2.5], loss_option="l2_norm", nb_levels=4, device="cpu",
                            scalar_last=False)
        try:
            mod = lm._PWCLONetLossModule(cfg, lm.Pose("quaternions"))
        except Exception:   # Pose(...) may need pyquaternion; build the module around the same forward
            mod = lm._PWCLONetLossModule.__new__(lm._PWCLONetLossModule)
            torch.nn.Module.__init__(mod)
            mod.config, mod.pose = cfg, None
            mod.exp_weighting = lm.ExponentialWeights(2, cfg.init_weights) if exp else None
            mod.weights = None if exp else cfg.loss_weights
            mod.degrees, mod.loss_config, mod.nb_levels = False, cfg.loss_option, 4
        pred, gt = inputs(seed, batch)
        pred.requires_grad_(True)
        loss, log = mod(pred, gt)
        loss.backward()
        tag = "s%d" % seed
        out[tag + ".loss"] = loss.detach().numpy()
        out[tag + ".grad_pred"] = pred.grad.numpy()
        if exp:
            out[tag + ".grad_s"] = mod.exp_weighting.s_param.grad.numpy()
        for k, v in log.items():
            out[tag + ".log." + k] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v)
        meta["cases"].append(dict(seed=seed, batch=batch, with_exp_weights=exp, loss_weights=[1.0, 0.5],
                                  init_weights=[0.0, -2.5], keys=sorted(log.keys())))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items() if k.endswith(".loss")})


if __name__ == "__main__":
    main()
