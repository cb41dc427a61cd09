"""Stock PointNet++ modules of the ``pointnet2_ops`` package (the package's callers of ball_query, three_nn and
three_interpolate) against outputs and gradients recorded from the
imported reference's modules (``tests/golden/modules_cases.npz``, written by ``oracle/gen_modules_golden.py`` in
the build container: reference Python on CPU, the C oracle standing in for its CUDA-only ``_ext``)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import module_cases as mc, params
from pwclonet_pylidarslam_amd.pointnet2_ops import pointnet2_modules, pointnet2_utils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "modules_cases.npz")


def _close(got, want, tol):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    scale = float(np.abs(want).max()) or 1.0
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, (err, scale)


def test_stock_module_state_dict_keys():
    """Runs without a GPU: same parameter / buffer names and shapes as the reference's modules."""
    z = np.load(GOLD)
    keys = json.loads(bytes(z["state_keys_json"]).decode())
    mods = mc.build(pointnet2_modules, pointnet2_utils)
    assert set(keys) == set(mods)
    for name, m in mods.items():
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys[name], name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sa_msg", "sa_all", "sa_xyz", "fp", "lfp", "votenet"])
def test_stock_module_eval_outputs(cuda, name):
    z = np.load(GOLD)
    m = params.fill_module_generic(mc.build(pointnet2_modules, pointnet2_utils)[name]).to(cuda).eval()
    x = {k: v.to(cuda) for k, v in mc.inputs().items()}
    with torch.no_grad():
        out = mc.run(name, m, x)
    for k, v in out.items():
        want = z[f"{name}.{k}"]
        if k == "new_xyz":                               # sampled coordinates: copies, bit for bit
            assert np.array_equal(v.cpu().numpy(), want), (name, k)
        else:
            _close(v.cpu().numpy(), want, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mc.GRAD_CASES))
def test_stock_module_train_backward(cuda, name):
    """Train-mode forward (batch-statistics BN) and backward through grouping / interpolation: every parameter
    gradient and the feature-input gradients against the reference's."""
    z = np.load(GOLD)
    m = params.fill_module_generic(mc.build(pointnet2_modules, pointnet2_utils)[name]).to(cuda).train()
    x = {k: v.to(cuda) for k, v in mc.inputs().items()}
    for k in mc.GRAD_INPUTS[name]:
        x[k].requires_grad_(True)
    f = mc.run(name, m, x)["features"]
    (f * f).sum().backward()
    _close(f.detach().cpu().numpy(), z[f"{name}.train.features"], 2e-5)
    for k, p in m.named_parameters():
        _close(p.grad.cpu().numpy(), z[f"{name}.grad.{k}"], 2e-4)
    for k in mc.GRAD_INPUTS[name]:
        _close(x[k].grad.cpu().numpy(), z[f"{name}.grad_in.{k}"], 2e-4)
