#!/usr/bin/env python3
"""Which torch (non-hand-written) ops cost GPU time in one training step?  torch.profiler over 3 eager steps at B=32, grouped
by operator name (self device time), hand-written kernels (library launches show up as their own kernel names) excluded."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pwclonet_pylidarslam_amd
pwclonet_pylidarslam_amd.configure_hw_queues(8)
import torch
import bench
from pwclonet_pylidarslam_amd.loss import PWCLONetLossModule
from pwclonet_pylidarslam_amd.pwclonet import PWCLONet
from pwclonet_pylidarslam_amd.training import PWCLONetWithLoss, TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(7)
net = PWCLONet(dict(num_input_channels=3, sequence_len=2, device=str(dev), scalar_last=False, log_mode="none")).to(dev).train()
unit = PWCLONetWithLoss(net, PWCLONetLossModule(dict(with_exp_weights=True, init_weights=[0.0, -2.5], loss_option="l2_norm",
                                                     nb_levels=4, scalar_last=False)).to(dev))
opt = torch.optim.Adam(unit.parameters(), lr=1e-4, fused=True)
x1, x2 = bench.make_batch(32, 8192, 2000, dev)
gt = torch.zeros(32, 7, device=dev); gt[:, 3] = 1.0
ts = TrainStep(unit, opt, x1, x2, gt)
for _ in range(3): ts.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): ts.step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 0: rows.append((t / 3e3, e.count // 3, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("self device time per step: %.2f ms over %d op kinds" % (tot, len(rows)))
for t, n, k in rows[:45]:
    print("%8.3f ms %5d  %s" % (t, n, k[:110]))
