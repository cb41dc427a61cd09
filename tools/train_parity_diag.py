#!/usr/bin/env python3
"""Diagnostic for the train-mode parity bound: who differs from whom, and by how much.
Runs the n1024_b2 train-mode case (tests/golden/train_n1024_b2.npz) on the GPU through (a) the HIP training kernels
twice (run-to-run), (b) torch's own conv / batch_norm ops, and prints every recorded gradient's error relative to
max|g| against the values recorded from the imported reference on CPU."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_train as T
from oracle import gen_golden
from oracle.gen_grad_golden import ground_truth
from pwclonet_pylidarslam_amd.pointnet2_ops import pointnet2_utils, pytorch_utils as pt

dev = torch.device("cuda:0")
z = np.load(os.path.join(T.GOLDEN, "train_n1024_b2.npz"))
meta = json.loads(str(z["meta"]))
x1, x2 = gen_golden.case_inputs("n1024_b2")
x1, x2, gt = x1.to(dev), x2.to(dev), ground_truth(2).to(dev)
pointnet2_utils.deterministic_grads(True)
runs = {}
runs["hip_a"] = T._step(T._unit(dev), x1, x2, gt)
runs["hip_b"] = T._step(T._unit(dev), x1, x2, gt)
for flags in (("stack0", dict(_USE_HIP_STACK=False)), ("conv0", dict(_USE_HIP_STACK=False, _USE_HIP_CONV="0")),
              ("bn0", dict(_USE_HIP_STACK=False, _USE_HIP_BN=False)),
              ("torch", dict(_USE_HIP_STACK=False, _USE_HIP_CONV="0", _USE_HIP_BN=False))):
    old = {k: getattr(pt, k) for k in flags[1]}
    for k, v in flags[1].items():
        setattr(pt, k, v)
    runs[flags[0]] = T._step(T._unit(dev), x1, x2, gt)
    for k, v in old.items():
        setattr(pt, k, v)
names = list(runs)
print("%-74s " % "gradient error / max|g| vs reference golden" + " ".join("%9s" % n for n in names))
for k in meta["params"]:
    ref = torch.from_numpy(z["grad." + k]).double()
    row = []
    for n in names:
        g = runs[n][2]["pwclonet." + k].cpu().double()
        row.append((g - ref).abs().max().item() / ref.abs().max().item())
    print("%-74s " % k + " ".join("%9.2e" % v for v in row))
torch.use_deterministic_algorithms(True, warn_only=True)
runs["hip_c"] = T._step(T._unit(dev), x1, x2, gt)
torch.use_deterministic_algorithms(False)
print("run-to-run per tensor (hip_a vs hip_b | hip_a vs hip_c[torch deterministic mode]), nonzero only:")
for k in runs["hip_a"][2]:
    a, b, c = runs["hip_a"][2][k], runs["hip_b"][2][k], runs["hip_c"][2][k]
    sc = max(a.abs().max().item(), 1e-30)
    d1, d2 = (a - b).abs().max().item() / sc, (a - c).abs().max().item() / sc
    if d1 > 0 or d2 > 0:
        print("  %-80s %.2e %.2e" % (k, d1, d2))
print("forward run-to-run: pose %.2e loss %.2e" % ((runs["hip_a"][1] - runs["hip_b"][1]).abs().max().item(), abs(runs["hip_a"][0].item() - runs["hip_b"][0].item())))
print("run-to-run (hip_a vs hip_b): worst %.2e" % max(
    (runs["hip_a"][2][k] - runs["hip_b"][2][k]).abs().max().item() / max(runs["hip_a"][2][k].abs().max().item(), 1e-30)
    for k in runs["hip_a"][2]))
print("pose: " + " ".join("%s %.2e" % (n, (runs[n][1].cpu().double() - torch.from_numpy(z["pose_params"]).double()).abs().max().item()) for n in names))
