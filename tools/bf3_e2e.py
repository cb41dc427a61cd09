#!/usr/bin/env python3
"""Developer tool: end-to-end pose error of the two stack-layer paths (fp32 MFMA, PWCLO_BF16X3=1) against the
CPU oracle on the bench's random-init weights and KITTI-shaped pairs."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pwclonet_pylidarslam_amd  # noqa: F401
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import torch

from oracle import model as omodel
from pwclonet_pylidarslam_amd import synthetic
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet

dev = torch.device("cuda:0")
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (the bench's own batch: 8 ray-cast scenes + jittered copies)
x1, x2 = (t.cpu() for t in bench.make_batch(pairs, 8192, 1000, dev))
poses = {}
for flag in ("0", "1"):
    os.environ["PWCLO_BF16X3"] = flag
    torch.manual_seed(1234)
    net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False)).to(dev).eval()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    net.prepare_fused()
    with torch.no_grad():
        poses[flag] = net(x1.to(dev), None, x2.to(dev), None)[0].cpu()
os.environ["PWCLO_BF16X3"] = "0"
want = omodel.pwclonet_forward(sd, x1, x2)
scale = want.abs().max()
for flag in ("0", "1"):
    per = (poses[flag] - want).abs().flatten(1).max(dim=1).values
    print(f"PWCLO_BF16X3={flag}: per-pair max |pose - oracle|: " + " ".join(f"{v:.1e}" for v in per.tolist()))
per = (poses["0"] - poses["1"]).abs().flatten(1).max(dim=1).values
print("between the two paths, per pair: " + " ".join(f"{v:.1e}" for v in per.tolist()))
for flag in ("0", "1"):
    d = (poses[flag] - want).abs().max()
    print(f"PWCLO_BF16X3={flag}: max |pose - oracle| = {d:.3e}  (relative to max |pose| {d / scale:.3e})")
print(f"between the two paths: {(poses['0'] - poses['1']).abs().max():.3e}; max |pose| {scale:.3f}")
