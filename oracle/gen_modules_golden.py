#!/usr/bin/env python3
"""Generate tests/golden/modules_cases.npz from the IMPORTED reference's stock PointNet++ modules (CPU, the C
oracle standing in for the CUDA-only ``_ext``) -- TEST INFRASTRUCTURE, build container only.

    python -m oracle.gen_modules_golden

Eval-mode outputs of every case of ``oracle.module_cases`` plus, for GRAD_CASES, a train-mode forward/backward
(batch-statistics BatchNorm): loss = sum(features^2), gradients of every parameter and of the feature inputs.
"""
import json
import os

import numpy as np
import torch

from oracle import module_cases as mc, params, ref_import

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "modules_cases.npz")


def main():
    ns = ref_import.load()
    torch.manual_seed(0)
    mods = mc.build(ns.pointnet2_modules, ns.pointnet2_utils)
    x = mc.inputs()
    out = {}
    keys = {}
    for name, m in mods.items():
        params.fill_module_generic(m)
        keys[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            for k, v in mc.run(name, m, x).items():
                out[f"{name}.{k}"] = v.numpy()
    for name in mc.GRAD_CASES:
        m = mods[name]
        params.fill_module_generic(m)
        m.train()
        xi = {k: (v.clone().requires_grad_(True) if k in mc.GRAD_INPUTS[name] else v) for k, v in x.items()}
        f = mc.run(name, m, xi)["features"]
        (f * f).sum().backward()
        out[f"{name}.train.features"] = f.detach().numpy()
        for k, p in m.named_parameters():
            out[f"{name}.grad.{k}"] = p.grad.numpy()
        for k in mc.GRAD_INPUTS[name]:
            out[f"{name}.grad_in.{k}"] = xi[k].grad.numpy()
    out["state_keys_json"] = np.frombuffer(json.dumps(keys, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), len(out), "arrays,", os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
