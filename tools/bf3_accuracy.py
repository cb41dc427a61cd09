#!/usr/bin/env python3
"""Developer tool: error of the fp32-MFMA and the bf16x3-split set-upconv stack against a float64 reference
(run twice: PWCLO_BF16X3=0 and =1).  Prints max and rms error relative to the output scale."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pwclonet_pylidarslam_amd
from pwclonet_pylidarslam_amd import fused
from pwclonet_pylidarslam_amd.pointnet2_ops.pointnet2_modules import PointnetFPModulePWCLONet
from oracle import ops as O

dev = torch.device("cuda:0")
torch.manual_seed(3)
mod = PointnetFPModulePWCLONet(nsample=8, mlp=[64, 128, 64], post_mlp=[64 + 32, 64], radius=0.2, knn=True,
                               use_xyz=True, bn=True).eval()
for m in mod.modules():                     # non-trivial BN statistics
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
B, N2, N1 = [int(v) for v in (sys.argv[1:4] or (4, 2048, 1024))]
g = torch.Generator().manual_seed(1)
xyz2, xyz1 = (torch.rand(B, N2, 3, generator=g) * 2 - 1) * 10, (torch.rand(B, N1, 3, generator=g) * 2 - 1) * 10
f1 = torch.randn(B, N1, 64, generator=g)
idx = O.knn_point_with_dist(8, xyz1, xyz2)[1]
# float64 reference of the pooled stack: max_k relu(W2 relu(W1 [feat | diff] + b1) + b2)
layers = list(mod.mlp)
(w1, b1), (w2, b2) = fused.fold_conv_bn(layers[0]), fused.fold_conv_bn(layers[1])
w1, b1, w2, b2 = w1.double(), b1.double(), w2.double(), b2.double()
gi = idx.long()
nb_f = torch.stack([f1[b][gi[b]] for b in range(B)]).double()                      # (B,N2,8,64)
nb_x = torch.stack([xyz1[b][gi[b]] for b in range(B)]).double()
diff = nb_x - xyz2.double().unsqueeze(2)
x = torch.cat((nb_f, diff), dim=-1)                                                # [feat(64) | diff(3)]
h1 = torch.relu(x @ w1.T + b1)
h2 = torch.relu(h1 @ w2.T + b2)
ref = h2.max(dim=2).values                                                          # (B,N2,64)
up = fused.FusedUpconvHoisted(mod.to(dev))
(pre,) = fused.run_linear_jobs(up.jobs(f1.to(dev)))
pooled = torch.empty((B, N2, 64), dtype=torch.float32, device=dev)
d_xyz2, d_xyz1, d_idx = xyz2.to(dev), xyz1.to(dev), idx.to(dev)      # keep the device copies alive over the launch
fused._lib.call("upconv_fused_h_kernel_wrapper", dev, B, N1, N2, 8, d_xyz2.data_ptr(), d_xyz1.data_ptr(),
                pre.data_ptr(), d_idx.data_ptr(), up.packed.data_ptr(), pooled.data_ptr())
torch.cuda.synchronize()
err = (pooled.cpu().double() - ref).abs()
scale = ref.abs().max().item()
print("PWCLO_BF16X3=%s: max err / scale = %.3e, rms err / scale = %.3e (scale %.3f)" % (
    os.environ.get("PWCLO_BF16X3", "0"), err.max().item() / scale, err.pow(2).mean().sqrt().item() / scale, scale))
bad = (err / scale) > 1e-4
if bad.any():
    print("bad fraction %.4f" % bad.float().mean().item())
    print("by channel block of 16:", [round(bad[..., 16*o:16*o+16].float().mean().item(), 3) for o in range(4)])
    print("by channel mod 16     :", [round(bad[..., c::16].float().mean().item(), 3) for c in range(16)])
    print("by query mod 16       :", [round(bad[:, q::16].float().mean().item(), 3) for q in range(16)])
    print("by cloud              :", [round(bad[b].float().mean().item(), 3) for b in range(B)])
    qbad = bad.any(dim=2)
    print("first bad queries cloud0:", qbad[0].nonzero().flatten()[:24].tolist())

