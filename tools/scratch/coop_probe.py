import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pwclonet_pylidarslam_amd.pointnet2_ops import _ext as E
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.cat((torch.randn(16, 96000, 2, generator=g) * 18.0, torch.rand(16, 96000, 1, generator=g) * 3.0), dim=2).contiguous().to(dev)
E.furthest_point_sampling(x, 64); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); out = E.furthest_point_sampling(x, 8192); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
dt = sorted(ts)[1]
print("%.2f ms  %.3f us/iter" % (dt * 1e3, dt / 8191 * 1e6))
