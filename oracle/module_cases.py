"""Cases for the stock PointNet++ modules of the ``pointnet2_ops`` package -- TEST INFRASTRUCTURE.

One description shared by the fixture generator (``oracle/gen_modules_golden.py``: the imported reference's
modules on CPU, build container only) and by the tests (the product's modules, GPU box): constructor calls by
name, seeded inputs, the closed-form parameter fill of ``oracle.params.fill_module_generic``.
"""
import torch

B, N = 2, 512


def inputs():
    g = torch.Generator().manual_seed(20260)
    u = lambda *s: torch.rand(*s, generator=g)
    return {
        "xyz": u(B, N, 3) * 4 - 2,                    # (B,N,3) in [-2,2]^3
        "feat6": u(B, 6, N) * 2 - 1,                  # (B,6,N)
        "known": u(B, 128, 3) * 4 - 2,
        "known_feat": u(B, 6, 128) * 2 - 1,
        "unknown_feat": u(B, 8, N) * 2 - 1,
        "xyz2": u(B, 256, 3) * 4 - 2,
        "feat2": u(B, 4, 256) * 2 - 1,
    }


def build(mods, utils):
    """``mods`` / ``utils`` = a pointnet2_modules / pointnet2_utils pair (reference or product)."""
    return {
        "sa_msg": mods.PointnetSAModuleMSG(npoint=128, radii=[0.6, 1.2], nsamples=[8, 16],
                                           mlps=[[6, 16, 32], [6, 16, 48]], bn=True, use_xyz=True),
        "sa_all": mods.PointnetSAModule(mlp=[6, 32, 64]),
        "sa_xyz": mods.PointnetSAModule(mlp=[0, 16, 32], npoint=64, radius=1.0, nsample=16),
        "fp": mods.PointnetFPModule(mlp=[6 + 8, 32, 16], bn=True),
        "lfp": mods.PointnetLFPModuleMSG(mlps=[[6, 16]], radii=[1.0], nsamples=[8], post_mlp=[16 + 4, 16], bn=True),
        "votenet": utils.QueryAndGroupVoteNet(0.9, 8, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True),
    }


def run(name, module, x):
    """Returns a dict of named output tensors of case ``name``."""
    if name == "sa_msg":
        new_xyz, f = module(x["xyz"], x["feat6"])
        return {"new_xyz": new_xyz, "features": f}
    if name == "sa_all":
        new_xyz, f = module(x["xyz"], x["feat6"])
        assert new_xyz is None
        return {"features": f}
    if name == "sa_xyz":
        new_xyz, f = module(x["xyz"], None)
        return {"new_xyz": new_xyz, "features": f}
    if name == "fp":
        return {"features": module(x["xyz"], x["known"], x["unknown_feat"], x["known_feat"])}
    if name == "lfp":
        return {"features": module(x["xyz2"], x["xyz"], x["feat2"], x["feat6"])}
    if name == "votenet":
        f, g = module(x["xyz"], x["xyz2"], x["feat6"])
        return {"features": f, "grouped_xyz": g}
    raise KeyError(name)


GRAD_CASES = ("sa_msg", "fp", "lfp")      # train mode: loss = sum(features^2); grads of every parameter + inputs
GRAD_INPUTS = {"sa_msg": ("feat6",), "fp": ("unknown_feat", "known_feat"), "lfp": ("feat2", "feat6")}
